"""Arch registry with the reference's contract (basicsr/models/archs/__init__.py:22-47 of the
reference): `define_network(opt)` pops `type`, looks the class up among the `*_arch` modules and
instantiates it with the remaining options."""
import importlib

_ARCH_MODULE_NAMES = ["ast_arch", "astv2_arch", "fpro_arch", "hint_arch"]
_arch_modules = [importlib.import_module(f"basicsr.models.archs.{n}") for n in _ARCH_MODULE_NAMES]


def dynamic_instantiation(modules, cls_type, opt):
    cls_ = None
    for module in modules:
        cls_ = getattr(module, cls_type, None)
        if cls_ is not None:
            break
    if cls_ is None:
        raise ValueError(f"{cls_type} is not found.")
    return cls_(**opt)


def define_network(opt):
    network_type = opt.pop("type")
    return dynamic_instantiation(_arch_modules, network_type, opt)
