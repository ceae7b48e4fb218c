"""Loading calibration batches for `inference.py --load` without executing anything from the file.

The reference writes them with `pickle.dump([x.cpu() for x in synthetic_data], f)` (examples/asr/quantization/
synthesize.py:103-104) and reads them back with a bare `pickle.load` (inference.py:95-96).  Here the same file goes
through an Unpickler that can only rebuild tensors."""
import collections
import pickle

import numpy as np
import torch


def _storage_from_bytes(b):
    """torch.storage._load_from_bytes, but through the weights-only loader (the stock one unpickles anything)."""
    import io
    return torch.load(io.BytesIO(b), map_location='cpu', weights_only=True)


class _TensorListUnpickler(pickle.Unpickler):
    """Unpickler for `pickle.dump([tensor, ...])` files (synthesize.py:103-104; read back by the reference with a bare
    pickle.load, inference.py:95-96).  Only the callables a CPU tensor's reduce uses are resolvable; anything else -
    i.e. any attempt to run code from the file - raises."""
    _ALLOWED = {('torch._utils', '_rebuild_tensor_v2'): lambda: torch._utils._rebuild_tensor_v2,
                ('torch._utils', '_rebuild_tensor'): lambda: torch._utils._rebuild_tensor,
                ('torch.storage', '_load_from_bytes'): lambda: _storage_from_bytes,
                ('collections', 'OrderedDict'): lambda: collections.OrderedDict}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return self._ALLOWED[(module, name)]()
        if module == 'torch' and (name.endswith('Storage') or isinstance(getattr(torch, name, None), torch.dtype)):
            return getattr(torch, name)
        raise pickle.UnpicklingError(
            f'{module}.{name} is not allowed in a calibration-data file: --load takes a pickled list of CPU tensors '
            '(synthesize.py), a torch.save()d list (.pt) or an .npz of arrays')


def load_synthetic(path):
    """Calibration batches [B, 64, T]: .npz of arrays, torch.save()d list, or the reference's pickled list of tensors."""
    if path.endswith('.npz'):
        d = np.load(path)                                         # allow_pickle=False
        return [torch.from_numpy(d[k]) for k in sorted(d.files)]
    try:
        data = torch.load(path, map_location='cpu', weights_only=True)   # torch.save()d list
    except Exception:
        with open(path, 'rb') as f:
            data = _TensorListUnpickler(f).load()
    data = list(data)
    if not data or not all(isinstance(t, torch.Tensor) for t in data):
        raise ValueError(f'{path}: expected a list of tensors')
    return [t.detach().cpu() for t in data]
