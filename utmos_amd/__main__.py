"""Command line front end: ``python -m utmos_amd <command> [options]``.

Commands: ``select`` (the MI355X greedy selection, same flags and output as the reference's
``utmos select``, utmos/__main__.py:17-47 dispatches it the same way) and ``version``.
``convert`` belongs to the reference's VCF tooling and is not part of this build; ``select`` reads
``.vcf[.gz]``, ``.jl`` and ``.npz`` inputs directly.
"""
import sys

from . import __version__


def _usage():
    return (f"Utmos v{__version__} (MI355X build)\n"
            "usage: python -m utmos_amd COMMAND [OPTIONS]\n\n"
            "  select    pick the samples that capture the most variants (GPU)\n"
            "  version   print the version and exit\n")


def _run_select(options):
    from .select import select_main
    select_main(options)
    return 0


def _run_version(_options):
    print(f"Utmos v{__version__}")
    return 0


def _run_convert(_options):
    sys.stderr.write("utmos_amd: `convert` is not provided; give the VCF / .jl / .npz files to `select`\n")
    return 1


COMMANDS = {"select": _run_select, "version": _run_version, "convert": _run_convert}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        sys.stderr.write(_usage())
        return 0
    command, options = argv[0], argv[1:]
    if command not in COMMANDS:
        sys.stderr.write(f"unknown command {command!r}\n\n" + _usage())
        return 2
    return COMMANDS[command](options)


if __name__ == "__main__":
    sys.exit(main())
