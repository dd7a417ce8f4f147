"""The FIT half of ``Read.tweak_signal_normalization`` (/root/reference/nadavca/read.py:83-93) for many reads ON
THE HOST, with scipy: per read, keep the events whose mean lies within 1 of the model's expected level, sort the
pairs by mean and fit FITPACK's smoothing spline (``scipy.interpolate.splrep(means, expected, s=len(means))``).
Since round 3 the batch path fits on the device (``nvk_spline_fit_dev``, csrc/splfit.h: under that filter and
that ``s`` FITPACK never gets past its first trial, the least-squares cubic); this module is what serves a read the
kernel reports as outside that case, the per-read ``Read.tweak_signal_normalization``, and
``estimate_snps_batch(spline_fit='host')`` — the cross-check of the kernel (~0.25 ms per read; the reads are
independent, so the fits are spread over worker processes: scipy holds the GIL inside the call)."""
import numpy as np


def fit_one(observed, levels):
    """-> (t, c) of the cubic smoothing spline, or None when fewer than 4 usable events remain."""
    from scipy import interpolate
    with np.errstate(invalid='ignore'):
        keep = np.abs(levels - observed) <= 1      # an empty event has mean NaN and drops out
    xs, ys = observed[keep], levels[keep]
    if xs.size < 4:
        return None
    order = np.lexsort((ys, xs))
    t, c, _ = interpolate.splrep(xs[order], ys[order], s=len(xs))
    return t, c[:len(t)]


def _fit_chunk(args):
    means, expected, off = args
    return [fit_one(means[off[j]:off[j + 1]], expected[off[j]:off[j + 1]]) for j in range(len(off) - 1)]


_POOL = {}


def _pool(workers):
    """A process pool started with ``spawn`` (the parent holds a HIP context: never fork it); kept for the life
    of the process, the workers import numpy and scipy only."""
    if workers not in _POOL:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        _POOL[workers] = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context('spawn'))
    return _POOL[workers]


class FitHandle:
    """Fits of one set of reads under way (``submit_fits``); ``result()`` -> (t, c, knot_off, fitted) as
    ``fit_splines`` returns them."""

    def __init__(self, n, jobs, futures, local, means, expected, ref_off, workers):
        self.n, self.jobs, self.futures, self.local = n, jobs, futures, local
        self._retry = (means, expected, ref_off, workers)

    def result(self):
        results = [None] * self.n
        try:
            for (part, _), fut in zip(self.jobs, self.futures):
                for j, r in zip(part, fut.result()):
                    results[j] = r
        except Exception as exc:   # e.g. a main module the workers cannot import (interactive session)
            import sys
            means, expected, ref_off, workers = self._retry
            sys.stderr.write('nadavca_amd.splinefit: worker pool unusable (%s); fitting in this process\n' % exc)
            bad = _POOL.pop(workers, None)
            if bad is not None:
                bad.shutdown(wait=False, cancel_futures=True)
            for part, _ in self.jobs:
                for j in part:
                    results[j] = fit_one(means[ref_off[j]:ref_off[j + 1]], expected[ref_off[j]:ref_off[j + 1]])
        for j in self.local:
            means, expected, ref_off, _ = self._retry
            results[j] = fit_one(means[ref_off[j]:ref_off[j + 1]], expected[ref_off[j]:ref_off[j + 1]])
        return pack_results(results)


def pack_results(results):
    """per-read (t, c) or None -> (t, c, knot_off, fitted) laid end to end"""
    fitted = np.array([r is not None for r in results], dtype=bool)
    lens = np.array([len(r[0]) if r is not None else 0 for r in results], dtype=np.int64)
    knot_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    t = np.concatenate([r[0] for r in results if r is not None]) if fitted.any() else np.zeros(0)
    c = np.concatenate([r[1] for r in results if r is not None]) if fitted.any() else np.zeros(0)
    return t, c, knot_off, fitted


def merge_host_fits(means, expected, ref_off, t8, c8, fit):
    """The device's fits (t8, c8: (n, 8); fit: 0 fitted, 1 no fit, 2 outside the kernel's case) with the reads of
    class 2 fitted here by FITPACK -> (t, c, knot_off, fitted) as ``fit_splines`` returns them."""
    results = []
    for j in range(len(fit)):
        if fit[j] == 0:
            results.append((t8[j], c8[j]))
        elif fit[j] == 1:
            results.append(None)
        else:
            results.append(fit_one(means[ref_off[j]:ref_off[j + 1]], expected[ref_off[j]:ref_off[j + 1]]))
    return pack_results(results)


def submit_fits(means, expected, ref_off, usable, workers=0):
    """Start the fits of the usable reads and return at once: -> FitHandle.  With ``workers`` > 1 the reads go to
    the worker processes in runs (the caller goes on — e.g. launches the kernels of the previous chunk of reads —
    and collects with ``result()``); otherwise they are fitted in this process when ``result()`` is called."""
    n = len(ref_off) - 1
    idx = np.nonzero(usable)[0]
    if not (workers and workers > 1 and idx.size >= 4 * workers):
        return FitHandle(n, [], [], list(idx), means, expected, ref_off, workers)
    parts = np.array_split(idx, workers * 2)
    jobs = []
    for part in parts:
        if part.size == 0:
            continue
        lo, hi = int(ref_off[part[0]]), int(ref_off[part[-1] + 1])    # contiguous range covering the part
        off = np.concatenate([[ref_off[j] - lo for j in part], [ref_off[part[-1] + 1] - lo]])
        # (parts are runs of consecutive usable reads only where nothing in between is unusable; otherwise
        # cut per read)
        if np.array_equal(part, np.arange(part[0], part[-1] + 1)):
            jobs.append((part, (means[lo:hi], expected[lo:hi], off)))
        else:
            for j in part:
                a, b = int(ref_off[j]), int(ref_off[j + 1])
                jobs.append((np.array([j]), (means[a:b], expected[a:b], np.array([0, b - a]))))
    try:
        pool = _pool(workers)
        futures = [pool.submit(_fit_chunk, j[1]) for j in jobs]
    except Exception:
        return FitHandle(n, [], [], list(idx), means, expected, ref_off, workers)
    return FitHandle(n, jobs, futures, [], means, expected, ref_off, workers)


def fit_splines(means, expected, ref_off, usable, workers=0):
    """means / expected: f64 (sum R,) per-event means and expected levels of all reads end to end, read j at
    [ref_off[j], ref_off[j+1]); usable bool (n,): reads to fit.  -> (t, c, knot_off, fitted bool (n,)): knots
    and coefficients of the fitted reads end to end (read j's at [knot_off[j], knot_off[j+1]), empty when not
    fitted).  ``workers`` > 1: that many processes."""
    return submit_fits(means, expected, ref_off, usable, workers).result()
