"""imagemosaicing_amd -- MI355X-native (gfx950, HIP) pairwise match + homography-warp hot path of
YuhuaXu/ImageMosaicing, behind the C ABI of include/mi355_mosaic.h.

This package is a thin ctypes binding of libmi355mosaic.so (hand-written HIP kernels + host C++).  It has
no CPU compute path: importing it needs the built library, creating a Context needs a gfx950 device.
The function names mirror the reference's own per-pair interface (Ransac2D, SelectMatchPairs,
ImageProjectionTransform, MosaicImagesRefined ...).
"""
from .capi import (  # noqa: F401
    Context, Mi355Error, lib_path, load_library, default_params, Params,
    SFPOINT, KEYPOINT, DMATCH, MATCHPAIR, PAIR_RESULT, CHIPINFO, IMAGE_TRANSFORM, FEATURE_HEADER, FEATURE_RECORD_BYTES, comm_unique_id, comm_available,
    mosaic_layout, blend_layout, pair_schedule, surf_pair_schedule, resample_by_overlap, write_match_pairs, load_match_pairs, write_match_pairs_txt,
    write_transforms, load_transforms, write_keypoints, load_keypoints, results_to_match_pairs, global_affine_align, select_connected,
    global_affine_align_results, select_connected_results, pair_moments_host, select_connected_moments, global_affine_align_moments, PAIR_MOMENTS, write_descriptors_xml, load_descriptors_xml,
)

__all__ = [n for n in dir() if not n.startswith("_")]
