"""Package defaults (mirrors /root/reference/nadavca/defaults.py:4-8).

The reference names ``default/10kmer_fact2.h5`` as its default model, but that file
is not shipped with it; the packaged and documented default is the
6-mer table, carried here as ``default/kmer_model.npz`` (converted once from the
reference's ``default/kmer_model.hdf5`` by oracle/convert_model.py).
"""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
KMER_MODEL_FILE = os.path.join(_HERE, 'default', 'kmer_model.npz')
CONFIG_FILE = os.path.join(_HERE, 'default', 'config.yaml')
BWA_EXECUTABLE = 'bwa'
GROUP_NAME = 'Analyses/Basecall_1D_000'
RENORM_ROUNDS = 3
