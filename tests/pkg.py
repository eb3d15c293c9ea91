"""Test helper: imports the product package (directory mygram-db_amd/) as `mygram_db_amd`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

mg = entry.load_package()
