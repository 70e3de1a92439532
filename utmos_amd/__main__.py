#!/usr/bin/env python3
"""utmos_amd entry point: `python -m utmos_amd select ...` (same CMD layout as utmos/__main__.py:17-47)."""
import argparse
import sys

from utmos_amd import __version__
from utmos_amd.select import select_main


def version(args):  # pylint: disable=unused-argument
    """Print the version"""
    print(f"Utmos v{__version__}")


def convert(args):  # pylint: disable=unused-argument
    """VCF conversion is outside this build's scope (the reference does it with scikit-allel)."""
    sys.stderr.write("utmos_amd: `convert` is not part of the MI355X build; `select` reads .vcf[.gz], .jl and .npz directly\n")
    sys.exit(1)


TOOLS = {"convert": convert, "select": select_main, "version": version}

USAGE = f"""\
Utmos v{__version__} - Maximum-coverage algorithm to select samples for validation and resequencing

    CMDs:
        select   Select samples (MI355X)
        version  Print the version
"""


def main():
    parser = argparse.ArgumentParser(prog="utmos", description=USAGE,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("cmd", metavar="CMD", choices=TOOLS.keys(), type=str, default=None, help="Command to execute")
    parser.add_argument("options", metavar="OPTIONS", nargs=argparse.REMAINDER, help="Options to pass to the command")
    if len(sys.argv) == 1:
        parser.print_help(sys.stderr)
        sys.exit()
    args = parser.parse_args()
    TOOLS[args.cmd](args.options)


if __name__ == "__main__":
    main()
