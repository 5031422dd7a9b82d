"""Compose / SetConfig / SaveRaw / UpdateRaw / RemoveSplit (contract of dance/transforms/misc.py:15-190)."""
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


@register_preprocessor("misc")
class UpdateRaw(BaseTransform):
    """Cut ``data.data.raw`` down to the genes the matrix has now, in their order (dance/transforms/misc.py:154-174) — after a gene
    filter that ran past ``SaveRaw``.  A ``DeviceArray`` raw matrix is subset on the device."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)

    def __call__(self, data):
        import types

        from ..data import AnnDataLite
        raw = getattr(data.data, "raw", None)
        if raw is None:
            raise AttributeError(f"Raw data attribute doesn't exist \n{data}"
                                 f"If you wish to update raw, save raw first.")
        self.logger.warning("RawData will change.")
        pos = raw.var.index.get_indexer(data.data.var_names)
        if (pos < 0).any():
            raise KeyError("some genes of the current matrix are not in .raw")
        x = AnnDataLite._take(raw.X, pos, 1)
        data.data.raw = types.SimpleNamespace(X=x, var=raw.var.iloc[pos], shape=tuple(x.shape))
        return data


@register_preprocessor("misc")
class RemoveSplit(BaseTransform):
    """Drop the cells of one split (dance/transforms/misc.py:177-190)."""

    _DISPLAY_ATTRS = ("split_name", )

    def __init__(self, *, split_name: str, **kwargs):
        super().__init__(**kwargs)
        self.split_name = split_name

    def __call__(self, data):
        self.logger.info(f"Popping split: {self.split_name!r}")
        data.pop(split_name=self.split_name)
