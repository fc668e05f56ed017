"""HBM bytes per launch of the example configuration's rebuilt kernels from two `rocprofv3 --pmc` passes over
tools/exp/z32_layers.py (FETCH_SIZE, WRITE_SIZE in their own runs; corrections as tools/pmc_traffic.py: KiB, 2 x FETCH on gfx950),
beside the algorithmic bytes of each layer at B = 768.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/zf -- python3 tools/exp/z32_layers.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/zw -- python3 tools/exp/z32_layers.py
    python3 tools/exp/z32_pmc.py gpurun_out/zf gpurun_out/zw > profiles/r06_z32ex_pmc_traffic.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmc_traffic import per_kernel
B = 768
T32, T64, T128 = B * 64 * 32 * 32 * 4, B * 32 * 64 * 64 * 4, B * 2 * 128 * 128 * 4
KEYS = {   # key -> (substring of the kernel name, algorithmic bytes, what moves)
    "wgrad1x1_stream": ("wgrad1x1_stream_kernel<4, 4, true", 3 * T32, "dy + its BatchNorm-backward partner + the layer input"),
    "conv1x1_stream_fwd": ("conv1x1_stream_kernel<4, 4, false", 2 * T32, "input + output"),
    "conv1x1_stream_dgrad": ("conv1x1_stream_kernel<4, 4, true", 4 * T32, "two input tensors + gate + output"),
    "conv1x1_bwd_fused": ("conv1x1_bwd_wide_stream_kernel<true>", 4 * T32, "dy + partner + the layer input + dx (the weight gradient reads the first three again)"),
    "wgrad_s2_thin_conv0": ("wgrad_s2_thin_stream_kernel<2, 2, true", 2 * T64 + T128, "dy + partner (32 channels) + the image"),
    "wgrad_s2_thin_up1": ("wgrad_s2_thin_stream_kernel<2, 2, false", T64 + T128, "d1 + the image-side gradient"),
    "conv_s2_thin": ("conv_s2_thin_stream_kernel<2, 2>", (2 * T128 + 3 * T64) // 2, "mean of the two forms the bench launches: first convolution (image + 32-channel output) and the data gradient of the last transposed convolution (+ the gate tensor)"),
    "convT_thin": ("convT_thin_stream_kernel<2>", T64 + T128, "32-channel input + image-side output"),
    "wgrad_wide1_3x3": ("wgrad_wide1_kernel<3, 64, true", 3 * T32, "dy + partner + the layer input"),
    "wgrad_wide1_4x4": ("wgrad_wide1_kernel<4, 32, true", 2 * T32 + T64, "dy + partner (64 channels) + the 32-channel input"),
    "conv3x3_wide_stream_fwd": ("conv3x3_wide_stream_kernel<false>", 2 * T32, "input + output (forward form)"),
    "conv3x3_wide_stream_dgrad": ("conv3x3_wide_stream_kernel<true>", 5 * T32, "as tools/exp/z32_layers.py launches it: dy, its BatchNorm-backward partner (also the residual: read twice), the gate (also the statistics operand) + output"),
    "conv_s2_wide_stream": ("conv_s2_wide_stream_kernel<true>", T64 + T32, "32-channel input + 64-channel output"),
    "convT_wide_stream_fwd": ("convT_wide_stream_kernel<false>", T32 + T64, "64-channel input + 32-channel output"),
    "convT_wide_stream_dgrad": ("convT_wide_stream_kernel<true>", 2 * T32 + 2 * T64, "two input tensors + gate + output"),
    "conv_wide_3x3": ("conv_wide_kernel<1, 9, 4>", 2 * T32, "input + output (forward form)"),
}
fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for key, (pat, alg, what) in KEYS.items():
    names = [n for n in fetch if pat in n]
    if not names:
        continue
    n = max(names, key=lambda k: fetch[k])
    hbm = int((2 * fetch[n] + write.get(n, 0.0)) * 1024)
    out[key] = {"kernel": n[:120], "launches": nf[n], "hbm_bytes_per_launch": hbm, "algorithmic_bytes": alg, "ratio": round(hbm / alg, 3),
                "algorithmic_bytes_are": what}
json.dump(out, sys.stdout, indent=1)
print()
