"""CPU oracle for the Sailfish quantification hot path -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (sailfish_amd) must never import this.
"""
