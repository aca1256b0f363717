"""Kept for the tests' import path: the numpy models of the incremental decode live in oracle/incremental_ref.py."""
from oracle.incremental_ref import *  # noqa: F401,F403
from oracle.incremental_ref import incremental_decode, incremental_decode_v3  # noqa: F401
