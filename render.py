#!/usr/bin/env python3
"""Drop-in for the reference's ``python render.py ...`` entry point, rendering on the MI355X."""
import sys

import bhr_amd  # noqa: F401  (registers the package that lives in black-hole-renderer_amd/)
from bhr_amd.cli import main

if __name__ == "__main__":
    sys.exit(main())
