"""Command line (mirrors /root/reference/bin/nadavca:8-53): ``nadavca-amd [-k MODEL] [-g GROUP] [-b BWA] [-c CONFIG]
REFERENCE READ_DIR {snp [-i] [-o OUT] | align [-o DIR] | meth -p PATTERN [-o OUT] [-r ROUNDS]}``."""
import argparse

from . import defaults


def build_parser():
    parser = argparse.ArgumentParser(prog='nadavca-amd', description=__doc__.split('\n')[0])
    parser.add_argument('-k', '--kmer_model', help='file with k-mer model to use', default=defaults.KMER_MODEL_FILE)
    parser.add_argument('-g', '--group_name', default=defaults.GROUP_NAME,
                        help='name of group in fast5 files containing basecall info (default: %(default)s)')
    parser.add_argument('reference', help='reference fasta file')
    parser.add_argument('read_basedir', help='base directory of fast5 files')
    parser.add_argument('-b', '--bwa_executable', default=defaults.BWA_EXECUTABLE,
                        help='command used to run bwa-mem; only used if bwapy is unavailable')
    parser.add_argument('-c', '--configuration', default=defaults.CONFIG_FILE,
                        help='config file with parameters for estimator/aligner')
    parser.add_argument('-t', '--threads', type=int, default=1,
                        help='accepted for compatibility (the reference parses and ignores it too)')
    sub = parser.add_subparsers(dest='command')
    snp = sub.add_parser('snp', help='estimate SNPs')
    snp.add_argument('-i', '--independent', action='store_true',
                     help='treat each read independently and output probabilities separately for each read')
    snp.add_argument('-o', '--output', help='output file (directory with --independent) for the posteriors '
                                            '(default: stdout)')
    snp.set_defaults(function='snp')
    align = sub.add_parser('align', help='align signal to reference')
    align.add_argument('-o', '--output', help='output directory for alignments')
    align.set_defaults(function='align')
    meth = sub.add_parser('meth', help='detect methylation')
    meth.add_argument('-o', '--output', help='output file for methylation scores (default: stdout)')
    meth.add_argument('-p', '--pattern', type=str, required=True, help='pattern to detect')
    meth.add_argument('-r', '--renorm_rounds', type=int, default=defaults.RENORM_ROUNDS,
                      help='number of alternating renorm and realign rounds')
    meth.set_defaults(function='meth')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not getattr(args, 'function', None):
        build_parser().print_usage()
        return 2
    if args.function == 'snp':
        from .writers import estimate_snps_command
        estimate_snps_command(args)
    elif args.function == 'align':
        from .writers import align_signal_command
        align_signal_command(args)
    else:
        from .detect_meth import detect_meth_command
        detect_meth_command(args)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
