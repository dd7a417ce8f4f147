"""Helpers shared by the estimator-level tests: rebuild the simulated reads stored in
tests/golden/estimator.npz and tests/golden/workflows.npz (written by oracle/make_golden_estimator.py and
oracle/make_golden_workflows.py from the reference's own Python layer) with this package's classes."""
import json
import os

import numpy as np

from conftest import GOLDEN


class EstimatorFixture:
    def __init__(self, name='estimator.npz'):
        z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.z = z
        self.config = json.loads(str(z['config']))
        self.genome = np.array(list(str(z['genome'])))
        self.n = int(z['n_reads'])
        self.specs = []
        for i in range(self.n):
            keys, vals = z['r%d_map_keys' % i], z['r%d_map_vals' % i]
            self.specs.append(dict(raw_signal=np.array(z['r%d_raw_signal' % i]),
                                   sequence=np.array(list(str(z['r%d_sequence' % i]))),
                                   sequence_to_signal_mapping={int(a): int(b) for a, b in zip(keys, vals)},
                                   base_mapping=np.array(z['r%d_base_mapping' % i]),
                                   reverse=bool(z['r%d_reverse' % i])))

    def reads(self, normalize=True, subset=None):
        from nadavca_amd import synthetic
        from nadavca_amd.read import Read
        specs = self.specs if subset is None else [self.specs[i] for i in subset]
        reads = synthetic.reads_from_specs(specs)
        if normalize:
            Read.normalize_reads(reads)
        return reads

    def aligner(self):
        from nadavca_amd import synthetic
        from nadavca_amd.alignment import ApproximateAligner
        return synthetic.make_synthetic_aligner(ApproximateAligner, self.genome)
