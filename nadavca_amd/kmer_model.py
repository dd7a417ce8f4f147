"""k-mer model loading (mirrors /root/reference/nadavca/kmer_model.py:6-29).

``KmerModel.load_from_hdf5`` keeps the reference's name; it also accepts the .npz re-encoding this
package ships (default/kmer_model.npz), since h5py is an optional dependency."""
import numpy as np

from .alphabet import alphabet, inv_alphabet
from .dtw import KmerModel


def kmer_to_id(kmer):
    result = 0
    for base in kmer:
        result = result * len(alphabet) + inv_alphabet[base]
    return result


def load_kmer_model(filename, context=None):
    if str(filename).endswith('.npz'):
        return KmerModel.load_from_npz(filename, context=context)
    import h5py  # optional dependency
    with h5py.File(filename, 'r') as file:
        central_position = int(file.attrs['central_pos'])
        table = file['model'][()]
    mean = np.zeros(len(table))
    sigma = np.zeros(len(table))
    k = None
    for kmer, m, s in table:
        kmer = kmer.decode('ascii')
        idx = kmer_to_id(kmer)
        mean[idx], sigma[idx], k = m, s, len(kmer)
    return KmerModel(k, central_position, len(alphabet), mean, sigma, context=context)


KmerModel.load_from_hdf5 = staticmethod(load_kmer_model)
