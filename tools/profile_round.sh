#!/bin/bash
# Everything profiles/<prefix>_* is condensed from (run on the MI355X box from the repository root):
#     gpurun --timeout 1500 -- 'bash tools/profile_round.sh gpurun_out/r3z'
#     python tools/summarize_profiles.py gpurun_out/r3z profiles/round3
# Counter passes are separate rocprofv3 runs (--pmc never together with a trace); every command is bounded.
set -u
RUN=${1:?run directory under gpurun_out/}
mkdir -p "$RUN"
export TMPDIR=/tmp
T="timeout 280"
B="python bench.py --no-cpu-baseline --no-extra --no-traffic"
$T rocprofv3 --kernel-trace --stats -d $RUN/trace -o r1 --output-format csv -- $B > $RUN/bench_traced.json 2>/dev/null
$T rocprofv3 --pmc FETCH_SIZE -d $RUN/pmc_fetch -o f --output-format csv -- $B --steps 6 --warmup 2 > /dev/null 2>&1
$T rocprofv3 --pmc WRITE_SIZE -d $RUN/pmc_write -o w --output-format csv -- $B --steps 6 --warmup 2 > /dev/null 2>&1
$T rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU \
    -d $RUN/pmc1 -o p1 --output-format csv -- $B --steps 4 --warmup 2 > /dev/null 2>&1
$T rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM \
    -d $RUN/pmc2 -o p2 --output-format csv -- $B --steps 4 --warmup 2 > /dev/null 2>&1
$T rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT \
    -d $RUN/pmc3 -o p3 --output-format csv -- $B --steps 4 --warmup 2 > /dev/null 2>&1
# executed fp64 work (round 5): wave-instructions by kind, the matrix cores' operations, lane-cycles of the vector unit
$T rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 \
    -d $RUN/pmc4 -o p4 --output-format csv -- $B --steps 4 --warmup 2 > /dev/null 2>&1
# the evaluator's MD route against the rebuild-everything route: counters per atom and kernel times (tools/md_counters.sh)
bash tools/md_counters.sh > $RUN/eval_counters.txt 2>&1
# the same trace over the whole default run: the sub-lines of the other BASELINE configurations (fit, lead-0, evaluator) included
$T rocprofv3 --kernel-trace --stats -d $RUN/trace_extra -o x --output-format csv -- python bench.py --no-cpu-baseline --no-traffic > $RUN/bench_traced_extra.json 2>/dev/null
$T python tools/bench_kernels.py > $RUN/kernels.json 2> $RUN/kernels.err
timeout 400 python bench.py --steps 20 --warmup 5 > $RUN/bench_default.json 2> $RUN/bench_default.err
cut -c1-200 $RUN/bench_default.json
