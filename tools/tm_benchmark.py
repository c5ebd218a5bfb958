"""tm_benchmark for the HIP backend (benchmark/tm_benchmark.cc): same flags, same loop (one warm-up run_graph, then
-r timed blocking runs, host buffer in -> host buffer out), same output line -- but on the synthetic QUANTISED models
(the reference's benchmark tmfiles carry no weights and tm_benchmark hard-codes fp32).

    python tools/tm_benchmark.py -r 50 -s 1 -p int8            # MobileNet-v1 int8, batch 1
    python tools/tm_benchmark.py -r 20 -s 8 -p uint8 -b 8      # YOLOv3-tiny uint8, 8 images

Times are what tm_benchmark times: H2D of the input, the graph, D2H of the outputs (`tamd_graph_run`).  Needs a GPU:
the product has no CPU path."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tengine_amd import capi, models, tm2  # noqa: E402

# -s indices of tm_benchmark.cc:248-300 that have a synthetic counterpart here
MODELS = {0: ("squeezenet_v1.1", "squeezenet_v1.1", "fp32"), 1: ("mobilenetv1", "mobilenet_v1", "int8"),
          5: ("resnet50", "resnet50", "int8"), 8: ("yolov3_tiny", "yolov3_tiny", "uint8"), 9: ("mssd", "mssd", "uint8")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-r", "--loops", type=int, default=50)
    ap.add_argument("-t", "--threads", type=int, default=1, help="accepted for compatibility; the device ignores it")
    ap.add_argument("-s", "--model", type=int, default=1, help="0 squeezenet 1 mobilenetv1 5 resnet50 8 yolov3_tiny 9 mssd")
    ap.add_argument("-d", "--device", default="HIP")
    ap.add_argument("-p", "--precision", default=None, choices=["fp32", "int8", "uint8"])
    ap.add_argument("-b", "--batch", type=int, default=1)
    ap.add_argument("-f", "--model-file", default=None, help="a tmfile (e.g. from tools/save_graph.py) instead of -s")
    a = ap.parse_args()
    if a.device.upper() != "HIP":
        raise SystemExit("this benchmark drives the HIP device only (use the reference's tm_benchmark for CPU)")
    print("Tengine benchmark:\n  loops:    %d\n  threads:  %d\n  device:   HIP" % (a.loops, a.threads))
    print("backend: %s" % capi.version(), flush=True)
    if a.model_file:
        name, b = os.path.basename(a.model_file), open(a.model_file, "rb").read()
        g = tm2.read_tm2(b)
        dtype = {tm2.DT_FP32: "fp32", tm2.DT_INT8: "int8", tm2.DT_UINT8: "uint8"}[g.tensors[g.nodes[g.input_nodes[0]].outputs[0]].dtype]
    else:
        if a.model not in MODELS:
            raise SystemExit("no synthetic model for -s %d (have %s)" % (a.model, sorted(MODELS)))
        name, key, dtype = MODELS[a.model]
        dtype = a.precision or dtype
        g = models.build(key, dtype, a.batch)           # the benchmark graph whole, classifier Softmax included (on the device for all three dtypes)
        b = tm2.write_tm2(g)
    dt = {"fp32": tm2.DT_FP32, "int8": tm2.DT_INT8, "uint8": tm2.DT_UINT8}[dtype]
    gr = capi.Graph(b, batch=a.batch, direct_dispatch=True)      # the plugin's default: the blocking run as one AQL pass
    x = models.synth_input(models.set_batch(g, a.batch), 1, dt)
    gr.set_input(x)
    gr.run()                                  # warm-up, as tm_benchmark does
    cost = []
    for _ in range(a.loops):
        t0 = time.perf_counter()
        gr.run()
        cost.append((time.perf_counter() - t0) * 1e3)
    gr.close()
    print("%20s  min = %7.2f ms   max = %7.2f ms   avg = %7.2f ms   (%s, batch %d, %.0f img/s at min)" % (
        name, min(cost), max(cost), sum(cost) / len(cost), dtype, a.batch, a.batch * 1e3 / min(cost)), flush=True)


if __name__ == "__main__":
    main()
