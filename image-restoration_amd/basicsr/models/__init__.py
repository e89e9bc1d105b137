"""`create_model` of the reference (basicsr/models/__init__.py:23-46): the class named by `model_type`."""


def create_model(opt):
    from basicsr.models import image_restoration_model as _m
    model_type = opt["model_type"]
    model_cls = getattr(_m, model_type, None)
    if model_cls is None:
        raise ValueError(f"Model {model_type} is not found.")
    return model_cls(opt)
