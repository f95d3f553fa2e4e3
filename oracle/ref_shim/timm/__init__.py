"""Minimal stand-in for timm 0.9.10 (not installed here): only the layers the reference imports on the MLA path,
restated from timm's published semantics. Used ONLY by oracle/ref_import.py."""


def create_model(*a, **k):
    raise RuntimeError("timm ViT backbones are never instantiated by MLA (SURVEY 2.1 #19)")
