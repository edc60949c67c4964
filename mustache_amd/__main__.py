"""`python -m mustache_amd ...` -- the single-sample loop caller's command line (same flags as the reference's `mustache`).
The two-sample caller is `python -m mustache_amd.diff_mustache ...`."""
import sys

from . import mustache as _cli

if __name__ == "__main__":
    sys.exit(_cli.main(sys.argv[1:]))
