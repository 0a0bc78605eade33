"""Writes library_type_vectors.json: the expectations of the reference's own unit tests for library-type
compatibility (tests/LibraryTypeTests.cpp: "Paired-end library types have proper compatibility", "Single-end
library types have proper compatibility", and the encode/decode scenario), as plain data.

The reference's tests are Catch scenarios that loop over a name -> LibraryFormat table and state, per
combination, whether compatibleHit must hold.  This script re-expresses those stated expectations (it does not
run the reference) and records every combination with the enum values spelled out, so the checks in tests/
need neither the reference nor Catch.  Enum values: include/LibraryFormat.hpp:7-9; mate status:
0 SINGLE_END, 1 PAIRED_END_LEFT, 2 PAIRED_END_RIGHT."""
import json
import os

SE, PE = 0, 1
SAME, AWAY, TOWARD, NONE = 0, 1, 2, 3
SA, AS, S, A, U = 0, 1, 2, 3, 4
# the table the three scenarios share (tests/LibraryTypeTests.cpp:5-16, 37-48, 88-99)
FM = {"U": (SE, NONE, U), "SF": (SE, NONE, S), "SR": (SE, NONE, A), "IU": (PE, TOWARD, U), "ISF": (PE, TOWARD, S),
      "ISR": (PE, TOWARD, A), "OU": (PE, AWAY, U), "OSF": (PE, AWAY, S), "OSR": (PE, AWAY, A), "MU": (PE, SAME, U),
      "MSF": (PE, SAME, S), "MSR": (PE, SAME, A)}

pairs = []
for en, e in FM.items():                                   # :52-75
    for on in ["ISF", "ISR", "OSF", "OSR", "MSF", "MSR"]:
        ok = (en == on or (en == "IU" and on in ("ISF", "ISR")) or (en == "OU" and on in ("OSF", "OSR"))
              or (en == "MU" and on in ("MSF", "MSR")))
        pairs.append({"expected_name": en, "observed_name": on, "expected": e, "observed": FM[on], "compatible": ok})

singles = []
LEFT, RIGHT, SINGLE = 1, 2, 0
for en, e in FM.items():                                   # :109-160
    for fwd in (True, False):
        for ms in (LEFT, RIGHT, SINGLE):
            _, eo, es = e
            if es == U:
                ok = True
            elif es == S and eo != SAME and ((fwd and ms == SINGLE) or (fwd and ms == LEFT) or (not fwd and ms == RIGHT)):
                ok = True
            elif es == A and eo != SAME and ((not fwd and ms == SINGLE) or (not fwd and ms == LEFT) or (fwd and ms == RIGHT)):
                ok = True
            elif eo == SAME and ((es == S and fwd) or (es == A and not fwd)):
                ok = True
            else:
                ok = False
            singles.append({"expected_name": en, "expected": e, "is_forward": fwd, "mate_status": ms, "compatible": ok})

ids = [{"name": n, "format": f, "id": (f[0] & 1) | ((f[1] & 3) << 1) | ((f[2] & 7) << 3)} for n, f in FM.items()]   # :89-98 of LibraryFormat.hpp

out = {"source": "expectations of kingsfordgroup/sailfish tests/LibraryTypeTests.cpp, re-expressed as data",
       "paired": pairs, "single": singles, "format_ids": ids}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "library_type_vectors.json"), "w"), indent=0)
print(len(pairs), "paired,", len(singles), "single-end vectors")
