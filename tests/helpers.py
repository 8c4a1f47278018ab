"""Shared test plumbing: build the same problem for the CPU oracle and for the HIP product.

The decks are the reference's own (inputs/*.in; key / value content, shipped as data of the package
in jaybenne_amd/decks because /root/reference does not exist on the GPU box); the oracle is built
by oracle/harness.py.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from jaybenne_amd.deck import DECK_DIR, load_deck  # noqa: E402,F401
from oracle.harness import make_oracle, oracle_params, run_oracle_cycles  # noqa: E402,F401
