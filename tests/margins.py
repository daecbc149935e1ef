"""How much of each parity tolerance the device actually uses.

`record(tag, tensor, got, want, rtol, atol)` is called by the reference-golden `-m gpu` tests next to the assertion that enforces
the same bound; it keeps, per (fixture, tensor), the worst absolute error, the worst relative error and the worst
|got - want| / (atol + rtol * |want|) ("used": 1.0 = at the bound).  At interpreter exit the table is written to
$CTK_MARGINS_OUT (default gpurun_out/parity_margins.txt, which travels back from the GPU box); the copy the judge reads is
profiles/r04_parity_margins.txt."""
import atexit
import os

import numpy as np

_rows = {}


def record(tag, tensor, got, want, rtol=0.0, atol=0.0):
    got, want = np.asarray(got, np.float64).reshape(-1), np.asarray(want, np.float64).reshape(-1)
    if got.size != want.size or got.size == 0:
        return
    err = np.abs(got - want)
    bound = atol + rtol * np.abs(want)
    with np.errstate(divide="ignore", invalid="ignore"):
        used = float(np.nanmax(np.where(bound > 0, err / bound, np.where(err > 0, np.inf, 0.0))))
        rel = float(np.nanmax(np.where(np.abs(want) > 1e-3 * np.abs(want).max(), err / np.abs(want), 0.0))) if np.abs(want).max() > 0 else 0.0
    key = (tag.split(" step ")[0], tensor)
    r = _rows.setdefault(key, dict(abs=0.0, rel=0.0, used=0.0, rtol=rtol, atol=atol, n=0))
    r["abs"], r["rel"], r["used"] = max(r["abs"], float(err.max())), max(r["rel"], rel), max(r["used"], used)
    r["rtol"], r["atol"] = rtol, max(r["atol"], atol)
    r["n"] += 1


def _dump():
    if not _rows:
        return
    path = os.environ.get("CTK_MARGINS_OUT", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_margins.txt"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write("# observed error of the HIP path against the reference-recorded fixtures (tests/golden/*.npz), worst over the recorded steps\n")
            f.write("# used = max |got - want| / (atol + rtol * |want|): 1.0 = at the asserted bound; rel counts elements above 1e-3 of the tensor's max\n")
            f.write(f"{'fixture':58s} {'tensor':10s} {'max abs':>10s} {'max rel':>10s} {'rtol':>8s} {'atol':>8s} {'used':>7s}\n")
            for (tag, tensor), r in sorted(_rows.items()):
                f.write(f"{tag:58s} {tensor:10s} {r['abs']:10.3e} {r['rel']:10.3e} {r['rtol']:8.1e} {r['atol']:8.1e} {r['used']:7.3f}\n")
    except OSError:
        pass


atexit.register(_dump)


def close(tag, tensor, got, want, rtol=0.0, atol=0.0):
    """record the margin, then enforce the bound"""
    record(tag, tensor, got, want, rtol=rtol, atol=atol)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol)
