cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc6; mkdir -p $O
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TD_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$n -o r -- python $R/bench.py --steps 32 --warmup 16 --min-seconds 0 --cpu-scans 0 --ref-scans 0 --groups 1 --secondary 0 --upload-scans 0 > /dev/null 2> $O/$n.err
  python $R/tools/pmc_summary.py $O/$n knn_batch_kernel >> $O/summary.txt
  rm -rf $O/$n
done
cat $O/summary.txt
