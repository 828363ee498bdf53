"""CPU oracle for the ALIKED + LightGlue + local-BA hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path
(`opencv-simpleslam_amd/`) may import this package.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` use it,
and only as the checker / the reported CPU baseline.

Parity status (see DESIGN.md section "Oracle pinning"):
  * `oracle.pose_ref`      - PINNED against the reference's own
                             `slam/core/pose_utils.py` (imported in the build
                             container, vectors in tests/golden/pose_utils.npz).
  * `oracle.ba_ref` problem assembly - PINNED against the reference's own
                             `slam/core/ba_utils.py::_core_ba` run against
                             recording stubs (tests/golden/ba_assembly.npz).
  * `oracle.ba_ref` residual/Jacobian arithmetic, `oracle.lightglue_ref`,
    `oracle.aliked_ref`    - PARITY UNPINNED: the arithmetic lives in
                             third-party wheels (pycolmap==3.10.0 / pyceres==2.3,
                             lightglue==0.0 = cvg/LightGlue HEAD, torchvision,
                             kornia) that are absent from /root/reference and
                             from this image, and the reference ships no golden
                             vectors for them.  These modules restate the
                             published algorithms and are anchored on the
                             reference's call sites.
"""
