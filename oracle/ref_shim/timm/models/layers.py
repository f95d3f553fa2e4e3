from timm.layers import DropPath, trunc_normal_, lecun_normal_  # noqa: F401
