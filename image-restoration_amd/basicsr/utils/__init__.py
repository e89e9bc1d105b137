from .img_util import imwrite, tensor2img

__all__ = ["tensor2img", "imwrite"]
