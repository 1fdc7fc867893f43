# per-step latency from the plain-C host (no interpreter): tests/c_abi_harness.c in its `latency` mode
cd $GRAFT_REPO_ROOT
python -c "
from tests import c_harness
print(c_harness.write_fixture('gpurun_out/fixture_latency.txt'))" > /dev/null
tests/_build/c_abi_harness gpurun_out/fixture_latency.txt 0 latency | grep "^latency"
ABO_PHASE_EVENTS=1 tests/_build/c_abi_harness gpurun_out/fixture_latency.txt 0 latency | grep "^latency"
