export TMPDIR=/tmp
python tools/bench_scripts/qab.py 30 96 256,128,512 > gpurun_out/r04c_qab.txt 2>&1; cat gpurun_out/r04c_qab.txt | tail -12
python -m pytest tests -m gpu -q -x -k "fullsize or multirank or chamfer or stop_flags or objfit or silsetup" 2>&1 | tail -8
( time python bench.py ) > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err; tail -c 4000 gpurun_out/r04c_bench.json; tail -3 gpurun_out/r04c_bench.err
