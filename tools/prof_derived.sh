R=$PWD; OUT=$R/gpurun_out/pmc_derived; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for G in "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "WriteUnitStalled VALUUtilization" "OccupancyPercent" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $R/bench.py --steps 8 --warmup 31 --no-cpu-baseline --no-extra > $OUT/g$i.log 2>&1
  echo "group $i ($G): rc=$?" >> $OUT/summary.txt
done
cd $R; python3 tools/pmc_summary.py $OUT; cat $OUT/summary.txt; grep -il "error\|invalid\|not found" $OUT/*.log | head
