"""Writes tests/golden/{tiny,holes}.json: the inputs of the two known-answer micro-cases of
SURVEY.md Appendix C.1/C.2 together with the outputs the surveyor observed from the reference's
own code (transcribed in tests/cases.py).  These are the only reference-produced vectors for
this path; nothing here runs or reads /root/reference.

    python tests/golden/make_micro_cases.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.cases import (GOLDEN_DIR, HOLES_EXPECTED, TINY_EXPECTED, holes_case, tiny_case,  # noqa: E402
                         workload_to_json)

PROV = ("inputs: SURVEY.md Appendix C.%d; expected: outputs of the reference's own headers observed by the surveyor "
        "(SURVEY.md Appendix C.%d 'Observed')")

for mk, exp, k in ((tiny_case, TINY_EXPECTED, 1), (holes_case, HOLES_EXPECTED, 2)):
    w = mk()
    with open(os.path.join(GOLDEN_DIR, w.name + ".json"), "w") as f:
        json.dump({"provenance": PROV % (k, k), "input": workload_to_json(w), "expected": exp}, f, indent=1)
    print("wrote", w.name)
