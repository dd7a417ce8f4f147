"""Where the time of ``estimate_snps_batch`` / ``align_signal_batch`` goes: the bench's api_* workloads (10 000 reads
on a 10 kb reference) with every device stage wrapped in a synchronised wall clock.
`python tools/snps_stages.py [N] [snps|align]`."""
import sys
import time
import collections

sys.path.insert(0, '.')
import numpy as np
import torch
from nadavca_amd import synthetic, dtw, _lib, device, readbatch, defaults
import nadavca_amd.estimate_snps  # noqa: F401 (the package re-exports the function under the same name)
ES = sys.modules['nadavca_amd.estimate_snps']
from nadavca_amd.align_signal import _load_config

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
what = sys.argv[2] if len(sys.argv) > 2 else 'snps'
model = synthetic.load_model_arrays()
ctx = _lib.default_context()
km = dtw.KmerModel(*model, context=ctx)
rb, aligner, genome = synthetic.make_read_batch(n, model, seed=1000, genome_length=10000)
cfg = dict(_load_config(defaults.CONFIG_FILE), tweak_signal_normalization=True)
acc = collections.OrderedDict()


def wrap(mod, name):
    fn = getattr(mod, name)

    def timed(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return out
    setattr(mod, name, timed)


for nm in ('normalize_groups_dev', 'refine_alignment_dev', 'expected_levels_dev', 'event_means_dev', 'spline_fit_dev',
           'splev_groups_dev', 'estimate_log_likelihoods_dev', 'consensus_accumulate_dev', 'posterior_segments_dev',
           'refine_renorm_loop_dev', 'linfit_rescale_dev', 'to_host'):
    wrap(device, nm)
wrap(readbatch, 'signal_alignments')
wrap(device.DeviceBatch, 'from_windows')
for it in range(3):
    acc.clear()
    torch.cuda.synchronize()
    t = time.perf_counter()
    if what == 'snps':
        ES.estimate_snps_batch(genome, rb, config=cfg, kmer_model=km, aligner=aligner)
    else:
        from nadavca_amd.align_signal import align_signal_batch
        align_signal_batch(None, rb, kmer_model=km, aligner=aligner)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t
print('total %.1f ms (%.0f reads/s)' % (tot * 1e3, n / tot))
for k, v in acc.items():
    print('  %-32s %7.2f ms' % (k, v * 1e3))
print('  %-32s %7.2f ms' % ('(everything else)', (tot - sum(acc.values())) * 1e3))
