"""MI355X-native (gfx950) Tacotron2 acoustic-model hot path behind the reference's Python API.

Host side mirrors PhucNguyenAH/tacotron2_subword's ``model.py`` / ``hparams.py`` /
``distributed.py`` surface; all arithmetic runs in the hand-written HIP library
``libt2amd.so`` (C ABI in include/t2amd.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"
