"""Kernel classes of the training step (bench.py --mode train under rocprofv3, cut at the markers by tools/summarize_trace.py): share of kernel time,
ms and launches per step.     python tools/train_kernel_classes.py profiles/r06_train_timed_region.csv [profiles/r05_train_timed_region.csv ...]"""
import collections
import csv
import re
import sys

CLASSES = collections.OrderedDict([
    ("MIOpen Winograd (forward + data gradients)", r"winograd|Winograd|sp3AsmConv|miopenSp3"),
    ("MIOpen implicit GEMM / xdlops (weight gradients, strided layers)", r"igemm|xdlops|Xdlops|gridwise|ck::"),
    ("MIOpen naive / direct convolutions", r"naive_conv|MIOpenConv|gcnAsmConv|miopenGcn"),
    ("rocBLAS / Tensile GEMMs (1x1 layers)", r"Cijk|rocblas|gemm"),
    ("BatchNorm forward + backward", r"[Bb]atch[Nn]orm|batch_norm"),
    ("this repository's HIP kernels (sweep gradients, warp, gates, up-sampler, depthwise)", r"dvmvs::"),
    ("optimizer (multi-tensor Adam)", r"multi_tensor|[Aa]dam"),
    ("ATen elementwise / reductions / copies (gradient accumulation of shared weights, losses)", r"at::native|elementwise|vectorized|reduce_kernel|copyBuffer|fillBuffer"),
])


def table(path):
    rows = [r for r in csv.reader(open(path)) if r and not r[0].startswith("#") and r[0] != "Name"]
    meta = [r for r in csv.reader(open(path)) if r and r[0].startswith("#")][-1]
    steps = int(meta[meta.index("steps") + 1])
    total = sum(float(r[2]) for r in rows)
    agg, calls = collections.Counter(), collections.Counter()
    for r in rows:
        for name, pattern in CLASSES.items():
            if re.search(pattern, r[0]):
                break
        else:
            name = "other"
        agg[name] += float(r[2])
        calls[name] += int(float(r[1]))
    print(f"\n{path}: {total / 1e6 / steps:.1f} ms of kernel time and {sum(calls.values()) / steps:.0f} launches per step ({steps} steps; under the profiler)")
    for name, t in agg.most_common():
        print(f"  {name:88s} {100 * t / total:5.1f} %  {t / 1e6 / steps:7.2f} ms  {calls[name] / steps:6.0f} launches")


for p in sys.argv[1:]:
    table(p)
