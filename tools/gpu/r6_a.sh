#!/bin/bash
# round 6, first GPU call: the new tests (envelope, threads, HDF5 device resume, perf guard), then the whole suite, then
# the default bench line (baseline of this round)
out=gpurun_out/r06_a
mkdir -p $out
ls /opt/conda/bin/python3.9 > $out/conda_probe.txt 2>&1; /opt/conda/bin/python3.9 -W ignore -c "import h5py, numpy; print('h5py', h5py.version.version, 'numpy', numpy.__version__)" >> $out/conda_probe.txt 2>&1
cat $out/conda_probe.txt
timeout -k 10 900 python -m pytest tests/test_hip_envelope.py tests/test_h5_interop_runner.py tests/test_zz_perf_guard.py -q -m gpu --durations=15 -s > $out/pytest_new.txt 2>&1; rc=$?
tail -40 $out/pytest_new.txt; echo "new tests rc=$rc"
[ $rc = 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=10 > $out/pytest_gpu.txt 2>&1; rc=$?
tail -18 $out/pytest_gpu.txt; echo "suite rc=$rc"
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; rc=$?
cat $out/bench_default.json | cut -c1-1500; echo "bench rc=$rc"
