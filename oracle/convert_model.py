"""One-off converter (build container only): the reference's packaged k-mer model
(/root/reference/nadavca/default/kmer_model.hdf5, MPL-2.0 data file from
nanoporetech/tombo, see /root/reference/LICENSE.md) -> nadavca_amd/default/kmer_model.npz.

Run with an interpreter that has h5py (here: /opt/conda/bin/python3.9).  Indexing
follows the reference loader, /root/reference/nadavca/kmer_model.py:6-29:
kmer id = base-4 number of the k-mer string, A=0 C=1 G=2 T=3.
"""
import sys
import h5py
import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/nadavca/default/kmer_model.hdf5"
dst = sys.argv[2] if len(sys.argv) > 2 else "nadavca_amd/default/kmer_model.npz"
inv = {"A": 0, "C": 1, "G": 2, "T": 3}
with h5py.File(src, "r") as f:
    central = int(f.attrs["central_pos"])
    table = f["model"][()]
k = len(table[0][0].decode("ascii"))
mean = np.zeros(len(table))
sigma = np.zeros(len(table))
for kmer, m, s in table:
    idx = 0
    for ch in kmer.decode("ascii"):
        idx = idx * 4 + inv[ch]
    mean[idx] = m
    sigma[idx] = s
np.savez(dst, k=np.int64(k), central_pos=np.int64(central), alphabet_size=np.int64(4), mean=mean, sigma=sigma)
print(k, central, len(mean), mean.min(), mean.max(), mean.mean(), mean.std(), sigma.min(), sigma.max())
