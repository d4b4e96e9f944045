"""The reference's result files (geosss/io.py): its scripts keep chains, log-densities and ESS tables as pickles -- plain
or gzip-compressed, protocol 2, optionally guarded by a `<file>.lock` directory -- e.g. the `{method: samples}`
dictionaries of scripts/curve_vMF.py:123-124 and scripts/bingham.py:60,107.  `dump` / `load` here read and write that
format, so results move between the two packages in both directions.  Device tensors (the samplers' `as_tensor=True`
output) are brought to the host and stored as numpy arrays, anywhere inside dicts / lists / tuples, so a file written here
never needs torch to be read.

    dump(obj, filename, gzip=False, lock=None, timeout=None)      geosss/io.py:7-62
    load(filename, gzip=False, lock=None, timeout=None)           geosss/io.py:65-133
"""
import contextlib
import gzip as _gzip
import os
import pickle
import time


@contextlib.contextmanager
def _guard(filename, lock, timeout):
    """The lock of the reference: a directory `<filename>.lock` made before and removed after the access; a second
    process polls every 10 ms until it can make it, for ever or -- with a positive `timeout` -- until IOError."""
    if lock is None:
        yield
        return
    path = filename + ".lock"
    give_up = time.time() + timeout if timeout is not None and timeout > 0 else None
    while True:
        try:
            os.mkdir(path)
            break
        except FileExistsError:
            if give_up is not None and time.time() > give_up:
                raise IOError("Failed to acquire Lock")
            time.sleep(0.01)
        except OSError:
            raise IOError("Failed to acquire Lock")
    try:
        yield
    finally:
        try:
            os.rmdir(path)
        except OSError:
            raise IOError(f"missing lockfile {path}")


def _to_host(obj):
    """torch tensors -> numpy arrays, through the containers the scripts use"""
    try:
        import torch
    except ImportError:                                  # pragma: no cover
        torch = None
    if torch is not None and isinstance(obj, torch.Tensor):
        return obj.detach().cpu().numpy()
    if isinstance(obj, dict):
        return {k: _to_host(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_to_host(v) for v in obj]
    if isinstance(obj, tuple) and type(obj) is tuple:
        return tuple(_to_host(v) for v in obj)
    return obj


def dump(this, filename, gzip=False, lock=None, timeout=None):
    """Pickle `this` (protocol 2, as the reference) to `filename` (`~` expanded), gzip-compressed if asked."""
    filename = os.path.expanduser(filename)
    with _guard(filename, lock, timeout):
        with (_gzip.GzipFile(filename, "wb") if gzip else open(filename, "wb")) as stream:
            pickle.dump(_to_host(this), stream, protocol=2)


def load(filename, gzip=False, lock=None, timeout=None):
    """The object pickled in `filename`; IOError for an unreadable or truncated file."""
    filename = os.path.expanduser(filename)
    with _guard(filename, lock, timeout):
        with (_gzip.GzipFile(filename, "rb") if gzip else open(filename, "rb")) as stream:
            try:
                return pickle.load(stream)
            except (pickle.UnpicklingError, EOFError, OSError) as e:
                raise IOError(f"Failed to unpickle file: {e}")
