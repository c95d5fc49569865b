tag=r03y; export TMPDIR=/tmp; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(mktemp -d /tmp/pmc.XXXX)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline ) > gpurun_out/${tag}_pmc_$c.log 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_pmc_$c.csv
done
python tools/pmc_summary.py "query_kernel<2, 2" gpurun_out/${tag}_pmc_FETCH_SIZE.csv gpurun_out/${tag}_pmc_WRITE_SIZE.csv > gpurun_out/${tag}_pmc_query_human.json
python tools/pmc_summary.py "query_kernel<1, 3" gpurun_out/${tag}_pmc_FETCH_SIZE.csv gpurun_out/${tag}_pmc_WRITE_SIZE.csv > gpurun_out/${tag}_pmc_query_object.json
cat gpurun_out/${tag}_pmc_query_human.json; rm -f gpurun_out/${tag}_pmc_*.csv
