"""Stand-ins for the third-party packages the reference's scripts import and this image does not have (SURVEY.md 8(f) n4,
Appendix D "Missing"): installed into sys.modules ONLY for names whose real import fails, so that `train.py`,
`relighting.py`, `eval_nvs.py` of an unmodified NJU-3DV/Relightable3DGaussian checkout run against this repo's drop-in
extension packages (`r3dg_rasterization`, `bvh_tracing`, `simple_knn`).  tools/run_reference.py is the launcher.

Functional stand-ins (small, plain numpy / PIL / torch) for what the training path really calls:
    plyfile      PlyData / PlyElement for single-element vertex files     scene/dataset_readers.py:125-161, gaussian_model.py:507-666
    imageio      imread / imwrite                                          scene/utils.py:40-99
    torchvision  utils.make_grid / save_image, transforms.Resize,          train.py:247,314-317, utils/camera_utils.py:38-58
                 transforms.functional.InterpolationMode
    kornia       filters.spatial_gradient (kornia 0.6.12: Sobel / 8,       utils/loss_utils.py (first_order_edge_aware_loss)
                 replicate padding)
Everything else that is only imported, never called on that path (dearpygui, pyexr, cv2, nvdiffrast, open3d, trimesh,
the model zoo under torchvision.models that lpipsPyTorch imports ...) becomes an auto-mock.  `tensorboard` is left
missing on purpose: utils/system_utils.py:20-26 handles its ImportError itself."""
import enum
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys
import types
from unittest import mock

import numpy as np
import torch

MOCK_ONLY = ("dearpygui", "pyexr", "cv2", "nvdiffrast", "open3d", "trimesh", "lpips", "torch_scatter", "matplotlib")


# ---------------------------------------------------------------------------------------------------------- plyfile
_PLY_TYPES = {"f4": "float", "f8": "double", "u1": "uchar", "i1": "char", "u2": "ushort", "i2": "short", "u4": "uint",
              "i4": "int"}
_PLY_NAMES = {v: k for k, v in _PLY_TYPES.items()}
_PLY_NAMES.update({"float32": "f4", "float64": "f8", "uint8": "u1", "int8": "i1", "uint16": "u2", "int16": "i2",
                   "uint32": "u4", "int32": "i4"})


class _PlyProperty:
    def __init__(self, name, code):
        self.name, self.val_dtype = name, code


class PlyElement:
    """One element of a PLY file backed by a numpy structured array (`element["x"]`, `.properties`, `.data`)."""

    def __init__(self, name, data):
        self.name, self.data = name, data
        self.properties = tuple(_PlyProperty(n, data.dtype[n].str[1:]) for n in data.dtype.names)

    @staticmethod
    def describe(data, name):
        return PlyElement(name, np.asarray(data))

    @property
    def count(self):
        return self.data.shape[0]

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return self.data.shape[0]


class PlyData:
    """Reader / writer of binary-little-endian (and ascii) PLY files with scalar properties -- what the reference writes."""

    def __init__(self, elements, text=False):
        self.elements, self.text = list(elements), text

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def write(self, path):
        with open(path, "wb") as fh:
            head = ["ply", "format %s 1.0" % ("ascii" if self.text else "binary_little_endian")]
            for e in self.elements:
                head.append("element %s %d" % (e.name, e.count))
                for p in e.properties:
                    head.append("property %s %s" % (_PLY_TYPES[p.val_dtype], p.name))
            head.append("end_header")
            fh.write(("\n".join(head) + "\n").encode("ascii"))
            for e in self.elements:
                if self.text:
                    np.savetxt(fh, np.stack([e.data[n] for n in e.data.dtype.names], 1), fmt="%.9g")
                else:
                    fh.write(e.data.astype(e.data.dtype.newbyteorder("<"), copy=False).tobytes())

    @staticmethod
    def read(path):
        with open(path, "rb") as fh:
            if fh.readline().strip() != b"ply":
                raise ValueError("%s is not a PLY file" % path)
            fmt, layout = None, []
            while True:
                line = fh.readline()
                if not line:
                    raise ValueError("%s: unterminated PLY header" % path)
                tok = line.decode("ascii").split()
                if not tok or tok[0] == "comment":
                    continue
                if tok[0] == "format":
                    fmt = tok[1]
                elif tok[0] == "element":
                    layout.append((tok[1], int(tok[2]), []))
                elif tok[0] == "property":
                    if tok[1] == "list":
                        raise ValueError("list properties are not supported by this stand-in")
                    layout[-1][2].append((tok[2], _PLY_NAMES[tok[1]]))
                elif tok[0] == "end_header":
                    break
            elements = []
            for name, count, props in layout:
                if fmt == "ascii":
                    rows = np.loadtxt(fh, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
                    data = np.empty(count, dtype=[(n, c) for n, c in props])
                    for i, (n, _c) in enumerate(props):
                        data[n] = rows[:, i]
                else:
                    order = "<" if fmt == "binary_little_endian" else ">"
                    dt = np.dtype([(n, order + c) for n, c in props])
                    data = np.frombuffer(fh.read(dt.itemsize * count), dtype=dt, count=count)
                    data = data.astype([(n, c) for n, c in props])
                elements.append(PlyElement(name, data))
        return PlyData(elements, text=fmt == "ascii")


def _plyfile_module():
    m = types.ModuleType("plyfile")
    m.PlyData, m.PlyElement = PlyData, PlyElement
    return {"plyfile": m}


# ---------------------------------------------------------------------------------------------------------- imageio
def _imageio_module():
    from PIL import Image

    def imread(path, mode=None, **_kw):
        img = Image.open(path)
        if mode is not None:
            img = img.convert(mode)
        return np.asarray(img)

    def imwrite(path, data, **_kw):
        Image.fromarray(np.asarray(data)).save(path)

    m = types.ModuleType("imageio")
    m.imread, m.imwrite, m.imsave = imread, imwrite, imwrite
    v2 = types.ModuleType("imageio.v2")
    v2.imread, v2.imwrite = imread, imwrite
    m.v2 = v2
    # scene/envmap.py:8 calls imageio.plugins.freeimage.download() at import (a downloader for the HDR codec)
    m.plugins = types.SimpleNamespace(freeimage=types.SimpleNamespace(download=lambda: None))
    m.__path__ = []
    return {"imageio": m, "imageio.v2": v2}


# ---------------------------------------------------------------------------------------------------------- torchvision
def make_grid(tensor, nrow=8, padding=2, pad_value=0.0, **_kw):
    """torchvision.utils.make_grid for a [B,C,H,W] batch (or a list of [C,H,W]): `nrow` images per row, `padding` pixels of
    `pad_value` around every cell, single-channel images repeated to three channels."""
    if isinstance(tensor, (list, tuple)):
        tensor = torch.stack(list(tensor), 0)
    if tensor.dim() == 2:
        tensor = tensor[None]
    if tensor.dim() == 3:
        tensor = tensor[None]
    if tensor.shape[1] == 1:
        tensor = tensor.repeat(1, 3, 1, 1)
    B, Cn, H, W = tensor.shape
    if B == 1:
        return tensor[0]
    xmaps = min(nrow, B)
    ymaps = -(-B // xmaps)
    ch, cw = H + padding, W + padding
    grid = tensor.new_full((Cn, ch * ymaps + padding, cw * xmaps + padding), pad_value)
    for k in range(B):
        y, x = divmod(k, xmaps)
        grid[:, y * ch + padding:y * ch + padding + H, x * cw + padding:x * cw + padding + W] = tensor[k]
    return grid


def save_image(tensor, fp, nrow=8, padding=2, **_kw):
    from PIL import Image
    grid = make_grid(tensor, nrow=nrow, padding=padding) if (isinstance(tensor, (list, tuple)) or tensor.dim() == 4) \
        else (tensor if tensor.dim() == 3 else tensor[None])
    arr = grid.detach().float().cpu().mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr[..., 0] if arr.shape[-1] == 1 else arr).save(fp)


class InterpolationMode(enum.Enum):
    NEAREST = "nearest"
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"


class Resize:
    """torchvision.transforms.Resize on [..., H, W] tensors through torch.nn.functional.interpolate."""

    def __init__(self, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=True):
        self.size, self.mode, self.antialias = size, interpolation, antialias

    def __call__(self, img):
        size = (self.size, self.size) if isinstance(self.size, int) else tuple(self.size)
        x = img[None] if img.dim() == 3 else img
        if tuple(x.shape[-2:]) == size:
            out = x.clone()
        elif self.mode == InterpolationMode.NEAREST:
            out = torch.nn.functional.interpolate(x, size=size, mode="nearest")
        else:
            out = torch.nn.functional.interpolate(x, size=size, mode=self.mode.value, align_corners=False,
                                                  antialias=bool(self.antialias))
        return out[0] if img.dim() == 3 else out


def _torchvision_modules():
    tv = types.ModuleType("torchvision")
    tv.__path__ = []
    utils = types.ModuleType("torchvision.utils")
    utils.make_grid, utils.save_image = make_grid, save_image
    tr = types.ModuleType("torchvision.transforms")
    tr.__path__ = []
    tr.Resize, tr.InterpolationMode = Resize, InterpolationMode
    fn = types.ModuleType("torchvision.transforms.functional")
    fn.InterpolationMode = InterpolationMode
    tr.functional = fn
    models = mock.MagicMock(name="torchvision.models")          # lpipsPyTorch/modules/networks.py: imported, built lazily
    models.__path__, models.__name__ = [], "torchvision.models"
    tv.utils, tv.transforms, tv.models = utils, tr, models
    return {"torchvision": tv, "torchvision.utils": utils, "torchvision.transforms": tr,
            "torchvision.transforms.functional": fn, "torchvision.models": models}


# ---------------------------------------------------------------------------------------------------------- kornia
def _kornia_modules():
    from relightable3dgaussian_amd import train_step

    def spatial_gradient(x, mode="sobel", order=1, normalized=True):
        if mode != "sobel" or order != 1 or not normalized:
            raise NotImplementedError("stand-in: kornia.filters.spatial_gradient(mode='sobel', order=1, normalized=True) only")
        return train_step.spatial_gradient(x)

    k = types.ModuleType("kornia")
    k.__path__ = []
    f = types.ModuleType("kornia.filters")
    f.spatial_gradient = spatial_gradient
    f.laplacian = mock.MagicMock(name="kornia.filters.laplacian")
    k.filters = f
    return {"kornia": k, "kornia.filters": f}


# ---------------------------------------------------------------------------------------------------------- auto-mocks
class _MockLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__, m.__spec__, m.__name__ = [], spec, spec.name
        return m

    def exec_module(self, module):
        pass


class _MockFinder(importlib.abc.MetaPathFinder):
    def __init__(self, roots):
        self.roots = set(roots)

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, _MockLoader(), is_package=True)
        return None


def _importable(name):
    if name in sys.modules:
        return True
    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
        return False


def install(verbose=False):
    """Install a stand-in for every package of the lists above that cannot be imported here.  Returns the names replaced
    (functional stand-ins first, auto-mocks after)."""
    replaced = []
    for root, factory in (("plyfile", _plyfile_module), ("imageio", _imageio_module), ("torchvision", _torchvision_modules),
                          ("kornia", _kornia_modules)):
        if not _importable(root):
            sys.modules.update(factory())
            replaced.append(root)
    mocked = [r for r in MOCK_ONLY if not _importable(r)]
    if mocked:
        sys.meta_path.append(_MockFinder(mocked))
    if verbose:
        print("[reference_shims] stand-ins: %s; import-only mocks: %s" % (", ".join(replaced) or "-", ", ".join(mocked) or "-"))
    return replaced + mocked
