"""Build driver: compiles the product libraries (hipcc, gfx950) and, for tests only, the oracle."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)


def _make(directory, *targets, quiet=True):
    cmd = ["make", "-C", directory] + (["-s"] if quiet else []) + list(targets)
    subprocess.run(cmd, check=True)


def build_product():
    """nbody_amd/lib/{libnbody_hip.so, libnbody.so, nbody-bench}; cross-compiles without a GPU."""
    _make(os.path.join(PKG_DIR, "csrc"), "all")


def build_oracle():
    """oracle/liboracle.so and, where /root/reference exists, oracle/_ref/* (the checker, not the product)."""
    _make(os.path.join(ROOT, "oracle"), "all")
    _make(os.path.join(ROOT, "oracle"), "ref-bench")
