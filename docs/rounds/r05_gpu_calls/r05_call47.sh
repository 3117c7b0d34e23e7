#!/bin/bash
# round 5, GPU call 47: the whole GPU suite on the final build; the few-waves configurations' profiles again
O=gpurun_out/r5_c47; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/pytest.log
tools/profile_config.sh r05_strong8_full --emulate-world 8 --record full > $O/p1.log 2>&1; echo "p1 rc $?"
tools/profile_config.sh r05_strong8_none --emulate-world 8 --record none > $O/p2.log 2>&1; echo "p2 rc $?"
tools/profile_config.sh r05_cfg2_full --rays 65536 --record full > $O/p3.log 2>&1; echo "p3 rc $?"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
