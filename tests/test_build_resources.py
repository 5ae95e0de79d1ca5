"""CPU suite: the hot kernels keep their registers.  hipcc reports the resource usage of every kernel at compile time; a
kernel that starts spilling to scratch keeps passing every numerical test and silently loses a factor in speed (round 3: a
template-parameter change in the shared epilogue made the prefill GEMM spill 528 bytes per lane -- 4x slower, found only by
chance).  This compiles the kernel sources with -Rpass-analysis=kernel-resource-usage and holds the kernels on the decode
step and on the prefill path to zero scratch."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "jukebox_amd", "csrc")
HOT = {
    "gemm.hip": ["gemm_glds_kernel", "gemm_8phase_kernel", "gemm_lds_kernelIDF16_", "gemm_lds_kernelIf", "gemm_kernelIDF16_Lb1", "gemm_kernelIfLb1", "gemv_lnf_kernelIDF16_Li1ELi8ELi8",
                 "gemv_lnf_kernelIDF16_Li1ELi16ELi10", "gemv_kernelIDF16_Li1ELi8ELb0ELb1ELi0", "gemv_kernelIfLi1ELi8ELb0ELb1ELi0",
                 "gemv_kernelIDF16_Li1ELi16ELb0ELb1ELi0", "gemv_merge_kernelILi1ELi8", "gemm_split_kernel", "gemm_split_glds_kernel"],
    "attention.hip": ["attn_decode_wide_kernelILi15", "attn_decode_mfma_kernelILi8", "attn_decode_mfma_kernelILi5ELb1",
                      "attn_decode_split_kernelILi8", "attn_prefill_v2_kernelILi30", "attn_prefill_v2_kernelILi16"],
}


@pytest.mark.parametrize("src", sorted(HOT))
def test_hot_kernels_do_not_spill(src, tmp_path):
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-c", os.path.join(CSRC, src),
                        "-o", str(tmp_path / "o.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    usage, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            usage[name] = int(m.group(1))
    for pat in HOT[src]:
        hits = {k: v for k, v in usage.items() if pat in k}
        assert hits, f"no kernel matching {pat} in {src}"
        assert all(v == 0 for v in hits.values()), f"{src}: scratch in {hits}"
