"""ORACLE -- CPU restatement of the reference's hot path (XiaRho/CMDA), used only as the checker.

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package; the
product (cmda_amd/) never does.  Parity status: PINNED -- every module here is checked against outputs of the
reference's own, unmodified modules (tests/golden/*.npz, generated in the authoring container by
tests/golden/make_golden.py, which imports /root/reference with import shims for the absent third-party packages).
The third-party pieces the reference relies on (mmcv ConvModule, timm DropPath, torch ops) are restated from their
documented behaviour; see SURVEY.md section 2e.
"""
