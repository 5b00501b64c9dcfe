"""ORACLE -- test infrastructure only.

CPU restatements of the reference's hot path (grouping in C / numpy, module stack in
plain PyTorch-CPU functional ops).  Nothing under frustum_convnet_amd/ may import this
package: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
only as the checker / the timed CPU baseline -- never as the thing shipped.

Pinning status (details in DESIGN.md):
* grouping.py / qdp_ref.c : the reference op is CUDA-only and unbuildable here; pinned by the
  mask criterion of the reference's ops/query_depth_point/test.py plus two independent
  restatements agreeing bit-for-bit.
* det_ref.py : pinned by golden vectors produced by importing the reference's own
  models/det_base.py on CPU in the build container (tests/golden/make_golden.py).
"""
