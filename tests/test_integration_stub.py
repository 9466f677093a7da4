"""The ctypes stub printed in INTEGRATION.md must match include/lmc_atomi.h (struct sizes) -- CPU only."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_stub_structs_match_the_binding():
    from lmc_atomi_amd import _capi
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n# lmc_hip.py.*?```", md, re.S).group(0)
    body = code.split("\n", 1)[1].rsplit("```", 1)[0]
    # keep only the two Structure definitions
    structs = re.findall(r"(class lmc_\w+\(C\.Structure\):\n(?:    .*\n|\s*\n)+?)(?=\n\S|\Z)", body)
    ns = {"C": C}
    for st in structs:
        exec(st, ns)
    assert C.sizeof(ns["lmc_problem"]) == C.sizeof(_capi.lmc_problem)
    assert C.sizeof(ns["lmc_myula_config"]) == C.sizeof(_capi.lmc_myula_config)
    assert [f[0] for f in ns["lmc_problem"]._fields_] == [f[0] for f in _capi.lmc_problem._fields_]
