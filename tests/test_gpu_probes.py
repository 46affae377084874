"""Two facts the kernels rely on, checked on the GPU the tests run on by stand-alone device programs (built by build()):
 * exp_neg (the device library's exp algorithm with its constants in scalar registers) equals exp bit for bit over 2^25
   arguments <= 0 and the special values -- it stands in for exp in the covariance functions and the normal density;
 * an element of the fp64 MFMA tile product is the k-ordered chain of fused multiply-adds from zero -- the speculated
   incumbent of the lattice route reproduces one element of the update's product that way, ahead of the update."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'probes')


@pytest.mark.parametrize('name', ['exp_neg_probe', 'mfma_chain_probe'])
def test_device_probe(name):
    exe = os.path.join(HERE, name)
    assert os.path.exists(exe), 'tests/probes/%s is not built: run __graft_entry__.build()' % name
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'differing 0' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
