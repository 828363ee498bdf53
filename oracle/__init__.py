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
  * `oracle.ba_ref.solve_dense_lm` - PARITY UNPINNED (pyceres absent): Ceres'
                             trust-region policy over one dense Jacobian; holds the
                             property the reference's own BA test pins.
  * `oracle.ransac_ref`    - PARITY UNPINNED (opencv_python==4.11.0.86 absent):
                             OpenCV's classic findFundamentalMat path restated.
  * `oracle.reproject_ref` - PINNED against the reference's own
                             `slam/core/pnp_utils.py::reproject_and_match_2d3d` run
                             under a cv2 stub (tests/golden/reproject_match.npz).
  The product-side `slam/core/trajectory_eval.py::sim3_align` is pinned the same way
  on the reference viewer's alignment (tests/golden/trajectory_alignment.npz).
"""
