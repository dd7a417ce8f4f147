"""Output writers and command entry points of the workflows (mirrors /root/reference/nadavca/align_signal.py:83-147
and /root/reference/nadavca/estimate_snps.py:72-127): the ``.npz`` per read that ``nadavca align``
writes and the TSV of posteriors that ``nadavca snp`` writes."""
import os
import sys

import numpy as np

from .estimator import Chunk


def fast5_files(directory):
    return [os.path.join(directory, f) for f in sorted(os.listdir(directory))
            if f.endswith('.fast5') and not os.path.isdir(os.path.join(directory, f))]


def labelled_raw_cut(read, apx_alignment, alignment):
    """The raw samples between the first and the last event start, and a per-sample base label: the reference
    base at every event start, 'N' elsewhere (align_signal.py:116-131).  Raises ValueError when event starts do
    not increase (the reference's "bad alignment")."""
    alignment = np.asarray(alignment)
    first = int(alignment[0][1])
    raw_cut = np.asarray(read.raw_signal)[first:int(alignment[-1][1])]
    labels = np.full(raw_cut.shape, 'N')
    starts = alignment[:-1, 1]
    if len(starts) > 1 and np.any(np.diff(starts) <= 0):
        raise ValueError('bad alignment: event starts do not increase')
    bases = np.asarray(apx_alignment.reference_part)[:len(starts)]
    inside = (starts - first >= 0) & (starts - first < labels.size)
    labels[(starts - first)[inside]] = bases[inside]
    return raw_cut, labels


def write_alignment_npz(path, read, apx_alignment, alignment):
    """``path``.npz with arr_0 = raw cut, arr_1 = base labels, arr_2 = (reference start, strand, contig, read
    sequence) as strings — the reference's layout (align_signal.py:83-84,133-134).  False for an empty cut."""
    raw_cut, labels = labelled_raw_cut(read, apx_alignment, alignment)
    if len(raw_cut) == 0:
        return False
    info = np.array((str(apx_alignment.reference_range[0]), '-' if apx_alignment.reverse_complement else '+',
                     str(apx_alignment.contig_name), ''.join(np.asarray(read.sequence).tolist())))
    np.savez(path + '.npz', raw_cut, labels, info)
    return True


def _ensure_directory(path):
    if not os.path.exists(path):
        os.makedirs(path)
    if not os.path.isdir(path):
        sys.stderr.write('Failed to create directory {} (maybe a file with that name exists?)\n'.format(path))
        return False
    return True


def align_signal_command(args, aligner=None):
    from .align_signal import align_signal
    files = fast5_files(args.read_basedir)
    if args.output and not _ensure_directory(args.output):
        return
    results = align_signal(reference_filename=args.reference, reads=files, config=args.configuration,
                           kmer_model=args.kmer_model, bwa_executable=args.bwa_executable,
                           group_name=args.group_name, aligner=aligner)
    for (read, res), filename in zip(results, files):
        if res is None:
            continue
        base = os.path.splitext(os.path.basename(filename))[0]
        target = os.path.join(args.output, base) if args.output else base
        if write_alignment_npz(target, read, *res):
            print('done', filename)
        else:
            print('empty raw cut', filename)


def write_chunks(chunks, reference, file):
    Chunk.print_head(file)
    for chunk in chunks:
        chunk.print(file, reference)


def estimate_snps_command(args, aligner=None):
    from .estimate_snps import estimate_snps
    from .genome import Genome
    try:
        reference = Genome.load_from_fasta(args.reference)[0].bases
    except FileNotFoundError:
        sys.stderr.write("failed to process: reference {} doesn't exist\n".format(args.reference))
        return
    files = fast5_files(args.read_basedir)
    chunks = estimate_snps(reference_filename=args.reference, reads=files, reference=reference,
                           config=args.configuration, kmer_model=args.kmer_model,
                           bwa_executable=args.bwa_executable, independent=args.independent,
                           group_name=args.group_name, aligner=aligner)
    if chunks is None:
        return
    if args.independent:
        if args.output and not _ensure_directory(args.output):
            return
        for chunk, filename in zip(chunks, files):
            if args.output:
                base = os.path.splitext(os.path.basename(filename))[0]
                with open(os.path.join(args.output, base + '.txt'), 'w') as f:
                    write_chunks([chunk], reference, f)
            else:
                write_chunks([chunk], reference, sys.stdout)
    elif args.output:
        with open(args.output, 'w') as f:
            write_chunks(chunks, reference, f)
    else:
        write_chunks(chunks, reference, sys.stdout)
