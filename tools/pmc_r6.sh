# round 6: request-level counters of the two camera launches (what bounds them is the number of memory requests, not HBM bytes).  usage: tools/pmc_r6.sh TAG
TAG=${1:-r06pmc}
export BENCH_ARGS="--steps 20 --warmup 5 --no-cpu-baseline --profile-run"
export NVBX_BENCH_MIN_MS=30
bash tools/gpu_pmc.sh $TAG/tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum
bash tools/gpu_pmc.sh $TAG/tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
bash tools/gpu_pmc.sh $TAG/sq SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
bash tools/gpu_pmc.sh $TAG/sq2 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
bash tools/gpu_pmc.sh $TAG/ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
find gpurun_out/$TAG -name "*counter_collection.csv" -size +20M -delete
