#!/bin/bash
for v in w8_24 w8_42; do echo "== $v"; timeout -k 10 120 tools/gemm_time_$v 1024 | head -1; done
echo "== base"; timeout -k 10 120 tools/gemm_time 1024 | head -1
python -c "import __graft_entry__ as g; g.smoke()"
