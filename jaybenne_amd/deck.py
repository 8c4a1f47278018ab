"""Parthenon-style input decks (``<block>`` headers, ``key = value``, ``#`` comments, ``&``
continuation), read the way the reference reads them.

The reference consumes its decks through Parthenon's ``ParameterInput`` (absent here); the call
surface used on the hot path is ``GetInteger / GetReal / GetString / GetBoolean`` and the
``GetOrAdd*`` variants (reference src/jaybenne/jaybenne.cpp:163-223, src/mcblock/mcblock.cpp:40-137).
``modify`` mirrors the override mechanism of the reference's regression harness
(tst/regression_test.py:85-145: keys of the form ``block/key``).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

# the reference's own input decks (inputs/*.in: key / value content only), shipped as data of the
# package: what BASELINE.json:configs name, what the regression harness and bench.py run
DECK_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decks")


class ParameterInput:
    def __init__(self, text: Optional[str] = None):
        self.blocks: Dict[str, Dict[str, str]] = {}
        if text is not None:
            self.load_string(text)

    # ------------------------------------------------------------------ parsing
    @classmethod
    def from_file(cls, path: str) -> "ParameterInput":
        with open(path, "r") as fh:
            return cls(fh.read())

    def load_string(self, text: str) -> None:
        block = None
        pending_key = None
        for raw in text.splitlines():
            line = raw.split("#", 1)[0].strip()
            if not line:
                continue
            if line.startswith("<") and line.endswith(">"):
                block = line[1:-1].strip()
                self.blocks.setdefault(block, {})
                pending_key = None
                continue
            if block is None:
                raise ValueError(f"deck line outside any <block>: {raw!r}")
            cont = line.endswith("&")
            if cont:
                line = line[:-1].rstrip()
            if pending_key is not None:
                self.blocks[block][pending_key] += line
            else:
                if "=" not in line:
                    raise ValueError(f"deck line without '=': {raw!r}")
                key, val = line.split("=", 1)
                pending_key = key.strip()
                self.blocks[block][pending_key] = val.strip()
            if not cont:
                pending_key = None

    def modify(self, overrides: Dict[str, object]) -> None:
        """``{"parthenon/mesh/nx1": 128}`` style overrides (last path element is the key)."""
        for path, val in overrides.items():
            block, key = path.rsplit("/", 1)
            self.blocks.setdefault(block, {})[key] = str(val)

    # ------------------------------------------------------------------ queries
    def DoesBlockExist(self, block: str) -> bool:
        return block in self.blocks

    def DoesParameterExist(self, block: str, key: str) -> bool:
        return key in self.blocks.get(block, {})

    def _get(self, block: str, key: str) -> str:
        try:
            return self.blocks[block][key]
        except KeyError:
            raise KeyError(f"Parameter name '{key}' not found in block '{block}'") from None

    def GetString(self, block: str, key: str) -> str:
        return self._get(block, key)

    def GetInteger(self, block: str, key: str) -> int:
        return int(float(self._get(block, key)))

    def GetReal(self, block: str, key: str) -> float:
        return float(self._get(block, key))

    def GetBoolean(self, block: str, key: str) -> bool:
        v = self._get(block, key).strip().lower()
        if v in ("true", "1"):
            return True
        if v in ("false", "0"):
            return False
        raise ValueError(f"cannot read '{v}' as a boolean ({block}/{key})")

    def _get_or_add(self, getter, block, key, default):
        if self.DoesParameterExist(block, key):
            return getter(block, key)
        if isinstance(default, bool):
            text = "true" if default else "false"
        elif isinstance(default, float):
            text = repr(default) if math.isfinite(default) else str(default)
        else:
            text = str(default)
        self.blocks.setdefault(block, {})[key] = text
        return default

    def GetOrAddInteger(self, block: str, key: str, default: int) -> int:
        return self._get_or_add(self.GetInteger, block, key, default)

    def GetOrAddReal(self, block: str, key: str, default: float) -> float:
        return self._get_or_add(self.GetReal, block, key, default)

    def GetOrAddString(self, block: str, key: str, default: str) -> str:
        return self._get_or_add(self.GetString, block, key, default)

    def GetOrAddBoolean(self, block: str, key: str, default: bool) -> bool:
        return self._get_or_add(self.GetBoolean, block, key, default)


def load_deck(name: str, overrides: Optional[Dict[str, object]] = None) -> ParameterInput:
    """One of the reference's decks by name (``stepdiff``, ``stepdiff_ddmc``, ``stepdiff_smr``,
    ``stepdiff_smr_ddmc``, ``stepdiff_smr_hybrid``, ``inf``, ``inf_stiff``) with the override
    mechanism of the reference's regression harness (tst/regression_test.py:85-145)."""
    pin = ParameterInput.from_file(os.path.join(DECK_DIR, name + ".in"))
    if overrides:
        pin.modify(overrides)
    return pin
