#!/usr/bin/env python3
"""Every member of the int8 GEMM kernel family on a list of layer shapes (runs ON THE GPU BOX): the table the plan-time
autotune decides from, printed -- microseconds per launch and TOP/s for each pinned member (TAMD_FORCE_GEMM).  Made
for the question DESIGN.md §7 leaves open: where and why the LDS-DMA kernel (conv_igemm2) loses to the simple one.

    python tools/bench_members.py [resnet50|mobilenet_v1|custom] [batch]
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_I8 SQ_LDS_BANK_CONFLICT -- \\
        python tools/bench_members.py custom 32            # separate --pmc run, no tracing (see tools/pmc_model.sh)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import conv_graph  # noqa: E402
from tengine_amd import capi, tm2  # noqa: E402

# cin, hw, cout, k, s, p  -- ResNet-50 / MobileNet-v1 GEMM shapes
SHAPES = {
    "resnet50": [(64, 56, 64, 1, 1, 0), (64, 56, 64, 3, 1, 1), (64, 56, 256, 1, 1, 0), (256, 56, 64, 1, 1, 0), (128, 28, 128, 3, 1, 1),
                 (128, 28, 512, 1, 1, 0), (512, 28, 128, 1, 1, 0), (256, 14, 256, 3, 1, 1), (256, 14, 1024, 1, 1, 0),
                 (1024, 14, 256, 1, 1, 0), (512, 7, 512, 3, 1, 1), (512, 7, 2048, 1, 1, 0), (2048, 7, 512, 1, 1, 0)],
    "mobilenet_v1": [(32, 112, 64, 1, 1, 0), (64, 56, 128, 1, 1, 0), (128, 56, 128, 1, 1, 0), (128, 28, 256, 1, 1, 0),
                     (256, 28, 256, 1, 1, 0), (256, 14, 512, 1, 1, 0), (512, 14, 512, 1, 1, 0), (512, 7, 1024, 1, 1, 0),
                     (1024, 7, 1024, 1, 1, 0)],
    "custom": [(64, 56, 64, 3, 1, 1), (256, 14, 256, 3, 1, 1)],
    "custom3": [(64, 56, 64, 3, 1, 1), (128, 28, 128, 3, 1, 1), (256, 14, 256, 3, 1, 1), (512, 7, 512, 3, 1, 1), (128, 56, 128, 3, 2, 1),
                (256, 28, 256, 3, 2, 1), (512, 14, 512, 3, 2, 1)],
    # shallow-K pointwise layers (pw_stream / pw_rows territory): ResNet-50 2a / 2c / 3c, MobileNet-v1 2_1 .. 3_2
    "pw": [(64, 56, 64, 1, 1, 0), (64, 56, 256, 1, 1, 0), (128, 28, 512, 1, 1, 0), (32, 112, 64, 1, 1, 0), (64, 56, 128, 1, 1, 0),
           (128, 56, 128, 1, 1, 0), (128, 28, 256, 1, 1, 0)],
}
MEMBERS = ["igemm0", "igemm1", "igemm2", "igemm3", "igemm4", "igemm5", "igemm6", "igemm7", "igemm8", "igemm9", "gemm_direct",
           "pw_stream", "pw_rows", "conv_igemm2", "conv_pgemm_i8<128x64", "conv_pgemm_i8<128x128", "conv_pgemm_i8<64x64", "conv_pgemm_i8<64x128"]
if os.environ.get("BENCH_MEMBERS"):
    MEMBERS = os.environ["BENCH_MEMBERS"].split(";")


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    os.environ["TAMD_AUTOTUNE"] = "0"
    print("%-34s" % "shape (batch %d)" % batch + "".join("%18s" % m for m in MEMBERS))
    for cin, hw, cout, k, s, p in SHAPES[which]:
        g, x = conv_graph(1, batch, cin, hw, hw, cout, k, s, p, 1, 0, True, 1)
        b = tm2.write_tm2(g)
        macs = None
        cells = []
        for m in MEMBERS:
            os.environ["TAMD_FORCE_GEMM"] = m
            try:
                gr = capi.Graph(b)
            finally:
                del os.environ["TAMD_FORCE_GEMM"]
            gr.set_input(x)
            gr.run()
            prof = gr.profile(20)
            conv = [q for q in prof if q["macs"] > 0][-1]
            gr.close()
            macs = conv["macs"]
            # the cell names the kernel that actually ran: a member that does not apply to the shape falls back to the
            # planner's default, which shows as a different name in its column
            kn = conv["kernel"].replace("conv_igemm_i8", "ig").replace("conv_igemm2_i8", "ig2").replace("conv_pgemm_i8", "pg").replace("_i8", "")
            cells.append("%18s" % ("%.1f %s" % (conv["ms"] * 1e3, kn[:11])))
        print("%-34s" % ("%dx%d^2 -> %d k%d s%d  %.0f MMAC" % (cin, hw, cout, k, s, macs / 1e6)) + "".join(cells))


if __name__ == "__main__":
    main()
