#!/bin/bash
# GPU session 3: the whole GPU suite after the host-layer rewrite (new tests: full-size parity, predict_next, 2-rank CLIs,
# cl_vae device generation)
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/s3; mkdir -p $O
python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; tail -40 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
