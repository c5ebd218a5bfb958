import sys, threading, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import oracle
from tengine_amd import capi, models, tm2
g = models.build("mobilenet_v1", "int8", 1)
b = tm2.write_tm2(g)
xs = [models.synth_input(g, 100 + i) for i in range(4)]
wants = [oracle.run_graph(g, x)[0] for x in xs]
errs = []
def work(i, rounds):
    try:
        for r in range(rounds):
            gr = capi.Graph(b)
            for _ in range(3):
                gr.set_input(xs[i]); got = gr.run()[0]
                if not np.array_equal(got.reshape(wants[i].shape), wants[i]): errs.append("thread %d round %d mismatch %d" % (i, r, np.count_nonzero(got.reshape(wants[i].shape) != wants[i])))
            gr.close()
    except Exception as e:
        errs.append("thread %d: %r" % (i, e))
for trial in range(4):
    ts = [threading.Thread(target=work, args=(i, 3)) for i in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    print("trial", trial, "errors so far:", errs[:6])
