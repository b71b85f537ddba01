"""Drop-in for the reference's compiled package ``pointnet2`` (only ``_ext`` lives here).

Put ``<repo>/sam6d_amd`` on sys.path and the reference's
``pointnet2_utils.py`` line ``import pointnet2._ext as _ext`` resolves to the gfx950 kernels.
"""
