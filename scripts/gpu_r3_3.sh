# round 3, call 3: full GPU suite (new parity tests: full-size .onnx voices, C++ PCM vs oracle, heavy-tailed weights, RNG
# indexing), mrf3_kernel with the loads interleaved between the MFMAs vs mrf2 / conv-by-conv, RCCL at world size 1 (debug)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.log
BQ="--no-extra --no-cpu-baseline --min-seconds 0.5"
for g in 2 3; do
  PIPER_HIP_MRF_GEN=$g timeout 300 python bench.py $BQ --steps 200 > $O/b1_g$g.json 2>> $O/err.log
  PIPER_HIP_MRF_GEN=$g PIPER_HIP_MRF2=2 timeout 300 python bench.py $BQ --batch 16 --steps 20 --warmup 3 > $O/b16_g$g.json 2>> $O/err.log
  PIPER_HIP_MRF_GEN=$g PIPER_HIP_MRF2=2 timeout 300 python bench.py $BQ --config 4 --steps 10 --warmup 3 > $O/b64_g$g.json 2>> $O/err.log
done
for ou in 1 2 3; do
  PIPER_HIP_MRF3_OU=$ou timeout 300 python bench.py $BQ --steps 200 > $O/b1_ou$ou.json 2>> $O/err.log
done
PIPER_HIP_MRF2=0 timeout 300 python bench.py $BQ --batch 16 --steps 20 --warmup 3 > $O/b16_conv.json 2>> $O/err.log
PIPER_HIP_MRF2=0 timeout 300 python bench.py $BQ --config 4 --steps 10 --warmup 3 > $O/b64_conv.json 2>> $O/err.log
PIPER_HIP_MRF2=2 timeout 300 python bench.py $BQ --preset high --batch 8 --steps 5 --warmup 2 > $O/h8_g3.json 2>> $O/err.log
PIPER_HIP_MRF2=0 timeout 300 python bench.py $BQ --preset high --batch 8 --steps 5 --warmup 2 > $O/h8_conv.json 2>> $O/err.log
timeout 300 python bench.py $BQ --preset high --steps 50 > $O/h1_g3.json 2>> $O/err.log
PIPER_HIP_MRF2=0 timeout 300 python bench.py $BQ --preset high --steps 50 > $O/h1_conv.json 2>> $O/err.log
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"
B="python $GRAFT_REPO_ROOT/bench.py --batch 16 --no-extra --no-cpu-baseline --no-roofline --min-seconds 0 --steps 3 --warmup 1"
(cd /tmp && PIPER_HIP_MRF2=2 timeout 600 rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- $B > /dev/null 2>&1)
python scripts/pmc_summary.py $O/pmc1 > $O/pmc1.txt 2>&1
python scripts/pmc_dump.py $O/pmc1 mrf >> $O/pmc1.txt 2>&1
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
PIPER_BENCH_DEBUG=1 PIPER_BENCH_DIST=1 timeout 300 python -X faulthandler bench.py --no-extra --no-cpu-baseline --no-roofline --steps 20 > $O/nccl_ws1.json 2> $O/nccl_ws1.err; echo "nccl ws1 rc=$?" >> $O/nccl_ws1.err
cat $O/pytest_gpu.log; tail -5 $O/err.log; grep -v amdgpu.ids $O/nccl_ws1.err | tail -25; head -c 600 $O/nccl_ws1.json; echo
python - <<'PY'
import json,glob,os
O="gpurun_out/r3c/"
for f in sorted(glob.glob(O+"*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"ERR",e); continue
    r=d.get("roofline") or {}
    ks={k:v for k,v in r.get("kernels",{}).items() if k.startswith("mrf")}
    print("%-16s ms %8.3f val %7.1fM hifigan %.3f ms %5.1f TF step %.3f | %s" % (os.path.basename(f), d["ms_per_step"], d["value"]/1e6,
          r.get("stage_ms",{}).get("hifigan",0), r.get("stage_tflops",{}).get("hifigan",0), r.get("step",{}).get("frac",0),
          " ".join("%s %.1fus %.1fTF" % (k.replace("_kernel",""), v["avg_launch_us"], v["tflops"]) for k,v in ks.items())))
PY
grep -n "mrf" $O/pmc1.txt | cut -c1-330
