G="TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum;TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum;TCC_EA0_RDREQ_GMI_32B TCC_EA0_RDREQ_IO_32B TCC_READ_SECTORS_sum TCC_BUBBLE_sum"
for cfg in vdsen2_20_bf16 vdsen2_20_fp32 dsen2_20_fp32; do
  bash tools/pmc_groups.sh r05_dram_$cfg "$G" bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --roofline-seconds 1 --sustain-seconds 0 --other-seconds 0 > gpurun_out/r05_dram_$cfg.txt 2>&1
  tail -3 gpurun_out/r05_dram_$cfg.txt
done
