#!/usr/bin/env python3
"""Drop-in for the reference's ``main.py`` command line (``-train`` / ``-process``), running the Hourglass
on an MI355X through libcgs_hip.so.  See INTEGRATION.md."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cgs_amd import cli  # noqa: E402

if __name__ == "__main__":
    cli.main()
