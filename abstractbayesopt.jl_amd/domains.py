"""ContinuousDomain (src/domains/ContinuousDomain.jl:16-29): box bounds with the reference's
validation.  Host-side value type; consumed by optimize_acquisition."""
import numpy as np


class ContinuousDomain:
    def __init__(self, lower, upper):
        lower = np.asarray(lower, dtype=np.float64).reshape(-1)
        upper = np.asarray(upper, dtype=np.float64).reshape(-1)
        if lower.shape[0] != upper.shape[0]:
            raise ValueError("lower and upper must have the same length")          # :24
        if np.any(lower > upper):
            raise ValueError("lower bounds must be less than or equal to upper bounds")  # :25
        self.lower, self.upper = lower, upper
        self.bounds = list(zip(lower.tolist(), upper.tolist()))
