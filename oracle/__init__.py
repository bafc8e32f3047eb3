"""CPU oracle for the DG-TTA hot path — TEST INFRASTRUCTURE ONLY.

A plain-PyTorch (CPU, fp32) restatement of the reference algorithm for the path
named by BASELINE.json `north_star`.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import this package; the product
package `dg_tta_amd` never does (its ops raise when the HIP library is absent).

Pinning: every function here is checked against the reference implementation
imported from /root/reference in the build container; the resulting vectors are
committed under tests/golden/ together with the generating script
(tests/golden/make_golden.py).  The reference itself ships no tests or golden
vectors (tests/__init__.py is empty), so these generated fixtures are the pin.
The nnUNet PlainConvUNet is third-party (dynamic-network-architectures==0.2,
absent from /root/reference); its arithmetic is restated from torch.nn layers
with the topology of the reference's plans.json.
"""
