"""fractal-image-compression_amd -- MI355X (gfx950) grey encode hot path of bvk_ss19.

Import as `fic_amd` (see fic_amd.py at the repo root).  Everything that computes a block
match lives in csrc/ (HIP) behind the C ABI of include/fic.h; this package is the host-side
mirror of the reference's interface plus the multi-GPU plumbing.
"""
from . import capi, synth, sharding
from .capi import FicError, declared_symbols, geometry, write_run_gray, decode_gray_run, decode_rgb_run, encode_rgb, write_run_rgb
from .host import Encoder, FractalCompression, RasterImage, encode_gray, encode_rgb_per_channel
from .sharding import ShardedEncoder, shard_spans, shard_planes, gather_records, pack_records, unpack_records

__all__ = ["capi", "synth", "sharding", "FicError", "declared_symbols", "geometry", "write_run_gray", "decode_gray_run", "decode_rgb_run", "encode_rgb", "write_run_rgb", "Encoder",
           "FractalCompression", "RasterImage", "encode_gray", "ShardedEncoder", "shard_spans", "shard_planes",
           "gather_records", "pack_records", "unpack_records"]
