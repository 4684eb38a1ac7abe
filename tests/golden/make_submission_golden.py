"""Golden vectors for the submission CSV (SURVEY.md §8f NEXT-4): run in the dev container, where
/root/reference exists.  Imports the REAL writer (/root/reference/submission.py:6-52 — stdlib only, so
no shim is needed), feeds it the rows below and commits the file it writes plus the rows themselves as
JSON.  tests/test_submission_golden.py replays the rows through esa_pose_estimation_amd.pipeline.SubmissionWriter
and compares the two files byte for byte.  Data only: nothing of the reference's source is stored.

    python tests/golden/make_submission_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import contextlib
import importlib.util
import io
import json
import os
import shutil
import tempfile

import numpy as np


def rows():
    """(filename, q, r, real, kind): kind says how the numbers are handed over — the CSV text depends on
    it (csv writes str(value): python float repr, numpy float64 / float32 shortest repr, ints as ints)."""
    rng = np.random.RandomState(20261005)
    out = []
    names = ["img%06d.jpg" % i for i in rng.permutation(40)]          # unsorted on purpose
    for i, name in enumerate(names):
        q = rng.randn(4)
        q /= np.linalg.norm(q)
        r = rng.randn(3) * [0.3, 0.3, 4.0] + [0, 0, 9.0]
        out.append((name, q.tolist(), r.tolist(), bool(i % 3 == 0), ["list", "f64", "f32"][i % 3]))
    out.append(("img000007.jpg", [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 10.0], False, "list"))   # duplicate filename
    out.append(("img000007.jpg", [0.5, 0.5, 0.5, 0.5], [1e-7, -2.5e5, 3.0], False, "f64"))  # stable order, exponents
    out.append(("a b,c.jpg", [1, 0, 0, 0], [0, 0, 1], True, "int"))                         # quoting, ints
    out.append(("", [float("nan"), 1.0, -0.0, 1e300], [float("inf"), 2.0, 3.0], True, "list"))
    return out


def cast(v, kind):
    if kind == "f64":
        return np.asarray(v, np.float64)
    if kind == "f32":
        return np.asarray(v, np.float32)
    if kind == "int":
        return [int(x) for x in v]
    return list(v)


def feed(writer, data):
    for name, q, r, real, kind in data:
        (writer.append_real_test if real else writer.append_test)(name, cast(q, kind), cast(r, kind))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.dirname(os.path.abspath(__file__)))
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("ref_submission", os.path.join(a.ref, "submission.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    data = rows()
    w = mod.SubmissionWriter()
    feed(w, data)
    with tempfile.TemporaryDirectory() as td:
        with contextlib.redirect_stdout(io.StringIO()):
            w.export(out_dir=td, suffix="golden")
        shutil.copy(os.path.join(td, "submission_golden.csv"), os.path.join(a.out, "submission_golden.csv"))
    # an empty writer and a test-set-only writer as well
    with tempfile.TemporaryDirectory() as td:
        with contextlib.redirect_stdout(io.StringIO()):
            mod.SubmissionWriter().export(out_dir=td, suffix="empty")
        shutil.copy(os.path.join(td, "submission_empty.csv"), os.path.join(a.out, "submission_empty.csv"))
    with open(os.path.join(a.out, "submission_rows.json"), "w") as f:
        json.dump([dict(filename=n, q=[repr(x) for x in q], r=[repr(x) for x in r], real=real, kind=k)
                   for n, q, r, real, k in data], f, indent=0)
    print("wrote submission_golden.csv, submission_empty.csv, submission_rows.json to", a.out)


if __name__ == "__main__":
    main()
