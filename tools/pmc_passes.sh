#!/bin/bash
# rocprofv3 counter passes over tools/pmc_workload.py, one counter set per run (gfx950: at most 8 SQ / 4 TCC / 2 GRBM
# counters per pass; FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).  PMC_WORKLOAD=<script> replaces the workload.  Output: gpurun_out/pmc_<tag>/ + a summary
# gpurun_out/pmc_<tag>.txt per pass.  usage: tools/pmc_passes.sh <tag-prefix> [pass ...]   (passes: fetch write inst lds l2)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
pre=$1; shift
passes=${*:-fetch write inst lds}
for p in $passes; do
  case $p in
    fetch) ctr="FETCH_SIZE";;
    write) ctr="WRITE_SIZE";;
    inst)  ctr="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";;
    lds)   ctr="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE";;
    l2)    ctr="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum";;
    wait)  ctr="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";;
    *) echo "unknown pass $p"; exit 1;;
  esac
  out=$R/gpurun_out/pmc_${pre}_$p
  rm -rf "$out"
  echo "== pass $p: $ctr"
  timeout -s KILL 240 rocprofv3 --pmc $ctr -d "$out" -o x --output-format csv -- python3 "$R/${PMC_WORKLOAD:-tools/pmc_workload.py}" > "$out.log" 2>&1 || { echo "pass $p failed"; tail -5 "$out.log"; exit 1; }
  python3 "$R/tools/pmc_sum.py" "$out" > "$R/gpurun_out/pmc_${pre}_$p.txt" || exit 1
  rm -rf "$out"
done
