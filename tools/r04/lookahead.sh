#!/bin/bash
# the id lookahead. Parity tests of the hint, the CLI with and without it, and the untimed region
# of a twitter-size step with the hint against without (tools/r04/slide_costs.py: wall time of set_batch + slide).
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "lookahead or csr or renumber" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -3
for key in twitter friendster; do
  SLIDE_COSTS_ONLY="binned-sweep tables" timeout 1500 python tools/r04/slide_costs.py $key gpurun_out/r04/r04_slide_${key}_lookahead.jsonl 2>&1 | tail -4
done
