export PYTHONPATH=$PWD TMPDIR=/tmp; mkdir -p gpurun_out
timeout 120 python tools/pmc_gemm_target.py --time 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05f_gemm_vs_hipblaslt.log
cd /tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
  tag=$(echo $pass | cut -d" " -f1)
  rm -rf /tmp/pmc_$tag; timeout 150 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$tag -- python $GRAFT_REPO_ROOT/tools/pmc_gemm_target.py > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<PY
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_8phase" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: (sum(v) / len(v), len(v)) for k, v in acc.items()})
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r05f_pmc_gemm_8phase.log
