# round 3, call 1: GPU suite on the pruned tree, the new one-invocation bench line, RCCL at world size 1, and counters
# for mrf2_kernel forced at batch (what limits it: issue stalls / waits / LDS conflicts / clock)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
PIPER_BENCH_DIST=1 timeout 300 python bench.py --no-extra --no-cpu-baseline --steps 20 > $O/bench_nccl_ws1.json 2> $O/nccl_ws1.err
PIPER_HIP_MRF2=2 timeout 300 python bench.py --config 4 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_b64_mrf2.json 2> $O/b64_mrf2.err
PIPER_HIP_MRF2=2 timeout 300 python bench.py --batch 16 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_b16_mrf2.json 2>> $O/b64_mrf2.err
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"
C2="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM"
B="python $GRAFT_REPO_ROOT/bench.py --batch 16 --no-extra --no-cpu-baseline --no-roofline --min-seconds 0 --steps 3 --warmup 1"
(cd /tmp && PIPER_HIP_MRF2=2 timeout 600 rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- $B > /dev/null 2>&1)
(cd /tmp && PIPER_HIP_MRF2=2 timeout 600 rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -- $B > /dev/null 2>&1)
(cd /tmp && PIPER_HIP_MRF2=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/st_b16_mrf2 -- $B > /dev/null 2>&1)
python scripts/pmc_summary.py $O/pmc1 > $O/pmc1.txt 2>&1
python scripts/pmc_dump.py $O/pmc2 > $O/pmc2.txt 2>&1
python scripts/pmc_dump.py $O/pmc1 mrf2 >> $O/pmc2.txt 2>&1
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pytest_gpu.log; tail -3 $O/bench.err $O/nccl_ws1.err $O/b64_mrf2.err
python - <<'PY'
import json
O="gpurun_out/r3a/"
def show(f):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); return None
    r=d.get("roofline") or {}
    print(f, "ms %.3f val %.1fM launches %s" % (d["ms_per_step"], d["value"]/1e6, d["config"]["kernel_launches_per_step"]),
          "top", r.get("kernel"), "frac %.3f" % r.get("frac",0), "ov_us %.2f" % r.get("event_pair_overhead_us",0),
          "step %.3f" % r.get("step",{}).get("frac",0), "hifigan TF %.1f" % r.get("stage_tflops",{}).get("hifigan",0), "spec", d.get("speculation"))
    return d
d=show("bench.json")
if d:
    for e in d.get("extra_configs",[]):
        r=e.get("roofline") or {}
        print("  leg", e.get("leg"), "%.1fs" % e.get("leg_seconds",0), e.get("error") or ("val %.4g %s ms %.3f" % (e.get("value",0), e.get("unit"), e.get("ms_per_step",0) or e.get("ms_per_call_mean",0))),
              "top", r.get("kernel"), "frac %.3f" % r.get("frac",0), "step %.3f" % r.get("step",{}).get("frac",0), "hifigan %.1f" % r.get("stage_tflops",{}).get("hifigan",0), e.get("speculation",""))
    print("  cpu", d.get("cpu_baseline",{}).get("value"), "stage_ms", d["roofline"]["stage_ms"])
    for k,v in sorted(d["roofline"]["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:12]:
        print("   %-50s n %5.1f us %7.2f raw %7.2f TF %6.1f" % (k, v["launches_per_step"], v["avg_launch_us"], v["avg_launch_us_event_pair"], v["tflops"]))
show("bench_nccl_ws1.json"); show("bench_b64_mrf2.json"); show("bench_b16_mrf2.json")
PY
grep -n "mrf2" $O/pmc1.txt $O/pmc2.txt | cut -c1-400
