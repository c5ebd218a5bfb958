"""Shipped launch plans: what the plan-time autotune chose for the BASELINE configurations on the box the committed evidence
was taken on (tengine_amd/plans/<model>_<dtype>_b<batch>.txt, the library's own TAMD_PLAN_CACHE format, written by
tools/make_plans.py).  Nothing here decides anything: a plan file only pre-answers the timing races (csrc/plan_cache.hip: plan
cache) -- a header that names another build or candidate list voids the whole file, every cached choice is re-checked for
applicability before it is used, and layers the file does not hold are timed as they always were."""
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
PLAN_DIR = os.path.join(HERE, "plans")


def path_for(model, dtype, batch):
    return os.path.join(PLAN_DIR, "%s_%s_b%d.txt" % (model, dtype, batch))


def seed(dst, model, dtype, batch):
    """Copy the shipped plan of (model, dtype, per-GPU batch) to `dst` (this job's TAMD_PLAN_CACHE file) unless that file
    already exists.  Returns the shipped file's name, or None when there is none (the job then times everything itself).
    The library may compile a batch as two graphs of half the batch (tamd_options.split_batch, csrc/graph_pair.hip): the shipped
    plan of batch / 2 is merged in where there is one (every key carries its batch, so the entries cannot collide)."""
    src = path_for(model, dtype, batch)
    if not os.path.isfile(src) or os.path.exists(dst):
        return None
    shutil.copyfile(src, dst)
    half = path_for(model, dtype, batch // 2) if batch >= 2 and batch % 2 == 0 else None
    if half and os.path.isfile(half):
        lines = open(half).read().splitlines()
        if lines and lines[0] == open(src).readline().rstrip("\n"):          # same build and candidate list, or the file is void anyway
            with open(dst, "a") as f:
                f.write("".join(l + "\n" for l in lines[1:] if "\t" in l))
    return os.path.relpath(src, os.path.dirname(HERE))
