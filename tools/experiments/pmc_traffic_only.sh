cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- python $R/bench.py --steps 32 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --secondary 0 --groups 1 > /dev/null 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- python $R/bench.py --steps 32 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --secondary 0 --groups 1 > /dev/null 2> $O/pmc_write.err
cd $R
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*_results.db" | head -1) $(find $O/pmc_write -name "*_results.db" | head -1) knn_batch_kernel > $O/knn_batch_traffic.json
cat $O/knn_batch_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
