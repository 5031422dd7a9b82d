"""Compose / SetConfig (contract of dance/transforms/misc.py:15-151)."""
from .base import BaseTransform
from ..registry import register_preprocessor


@register_preprocessor("misc")
class Compose(BaseTransform):
    """Apply transforms in order; each mutates ``data`` in place (return values are ignored, misc.py:68-71)."""

    def __init__(self, *transforms, use_master_log_level: bool = True, **kwargs):
        super().__init__(**kwargs)
        failed = [t for t in transforms if not isinstance(t, BaseTransform)]
        if failed:
            raise TypeError(f"Expect all transform objects to be inherited from BaseTransform, found {failed}")
        self.transforms = transforms
        if use_master_log_level:
            for t in transforms:
                t.log_level = self.log_level
                t.logger.setLevel(self.log_level)

    def __repr__(self):
        return "Compose(\n" + "".join(f"  {t!r},\n" for t in self.transforms) + ")"

    def __getitem__(self, idx):
        return self.transforms[idx]

    def hexdigest(self) -> str:
        import hashlib
        return hashlib.md5("".join(t.hexdigest() for t in self.transforms).encode()).hexdigest()

    def __call__(self, data):
        for t in self.transforms:
            t(data)


@register_preprocessor("misc")
class SetConfig(BaseTransform):
    """Set the dance data config (which channels are features / labels)."""

    def __init__(self, config_dict, **kwargs):
        super().__init__(**kwargs)
        self.config_dict = config_dict

    def __call__(self, data):
        data.set_config_from_dict(self.config_dict, overwrite=True)


@register_preprocessor("misc")
class SaveRaw(BaseTransform):
    """Save the current matrix (and var) to ``data.data.raw`` (dance/transforms/misc.py:126-151).  A ``DeviceArray`` X is cloned
    on the device: ``raw.X`` stays there until host code reads it."""

    def __init__(self, exist_ok: bool = False, **kwargs):
        super().__init__(**kwargs)
        self.exist_ok = exist_ok

    def __call__(self, data):
        import copy
        import types

        from ..data import DeviceArray
        self.logger.info("Saving data to ``.raw``")
        if getattr(data.data, "raw", None) is not None:
            if self.exist_ok:
                self.logger.warning("Overwriting raw content...")
            else:
                raise AttributeError(f"Raw data attribute already exist and cannot be overwritten.\n{data}"
                                     f"If you wish to overwrite, set 'exist_ok' to True.")
        x = data.data.X
        x = DeviceArray(x.tensor.clone()) if isinstance(x, DeviceArray) else copy.deepcopy(x)
        data.data.raw = types.SimpleNamespace(X=x, var=data.data.var.copy(), shape=tuple(x.shape))
        return data
