"""Moved next to the oracle (oracle/gate_excess.py) so that __graft_entry__.smoke() does not import from tests/;
this name stays for the test modules."""
from oracle.gate_excess import *   # noqa: F401,F403
from oracle.gate_excess import TOL, ROW_BOUND, _record, check_pixels, check_rows   # noqa: F401
