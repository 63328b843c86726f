"""Output / checkpoint layout and chunk driver of quflow (quflow/simulation.py), for device runs.

`Simulation` keeps the on-disk layout of the reference's `QuSimulation` (simulation.py:357-478):

    <datapath>mat    (n, [k,] N, N) complex   one row per output chunk, row 0 = initial state
    <datapath>shr    (n, N^2) float           (optional qutypes: 'shr', 'shc')
    <datapath>time   (n,) float64             accumulated delta_time
    <datapath>step   (n,) int                 accumulated delta_steps
    <datapath><logger name>  (n, ...)         value of logger(W) per row
    <datapath>tol_auto / iterations / number_of_maxit  (n,) float64   the stepper's `stats` per chunk
    attributes of <datapath>: version, created, qutypes (pickled), loggers (pickled), N, info, prerun
    attributes of <datapath>args/: the solve arguments (`sim['dt'] = ...`; callables pickled)

every dataset resizable along axis 0, a call `sim(W, delta_time, delta_steps, **stats)` appends one
row to each (the callback protocol of `solve`, simulation.py:795-798), and `sim['mat', -1]`,
`sim['time']`, `sim['dt']` read back (:237-276).  Two storage backends behind one small interface:
h5py (`.h5` / `.hdf5` file names; the reference's own format, so a device run can append to and
resume from a file the reference wrote -- only when h5py is importable) and a plain directory of
appendable raw arrays + JSON metadata (any other name; no dependency).

`solve` is the reference's chunk loop (simulation.py:604-798): `steps_out` steps per
`integrator(W, dt, steps=...)` call, stats / time / hamiltonian kwargs as the reference passes them,
callbacks after every chunk.  With the default stepper (quflow_amd.isomp, no host hooks) the
trajectory stays RESIDENT on the device between the chunks (DeviceTrajectory: each chunk is one
qf_isomp call, exactly what the stepper would do with a host array, so results are bit-identical)
and only what the output needs comes down.  Restarting from `Simulation(filename)` continues
bit-identically (tests/test_simulation.py:130-168: the stepper restarts its iteration vector at every
chunk anyway, isospectral.py:430).

Function-space outputs ('fun', 'funL2': transforms.py) are outside the scope of this package.
"""
import base64
import datetime
import inspect
import json
import os
import pickle
import shutil
import warnings

import numpy as np

from .geometry import hbar

_default_qutypes = {'mat': None}
_qutype2varname = {'mat': 'mat', 'shr': 'shr', 'shc': 'shc'}
_pickled_argnames = ['qutypes', 'hamiltonian', 'forcing', 'integrator', 'callback', 'integrator_callback',
                     'strang_splitting']          # simulation.py:53
_info_args = ['info']
_stats_fields = ('tol_auto', 'iterations', 'number_of_maxit')      # simulation.py:409-412


class _ReferenceUnpickler(pickle.Unpickler):
    """Pickled callables in a file the reference wrote name `quflow.*` objects (`sim['hamiltonian'] =
    qf.solve_poisson` is stored as the global `quflow.laplacian.cpu solve_poisson`, simulation.py:203-217).
    A resume through this package maps each to its counterpart here (`quflow_amd.laplacian.solve_poisson`, ...):
    that is the reference's own runfile rule -- a stored `qf.solve_poisson` + `qf.isomp` selects the accelerated
    pair (simulation.py:554-562).  A `quflow.*` name without a counterpart falls through to the ordinary
    import (and fails with the ordinary error when quflow is not installed)."""

    def find_class(self, module, name):
        if module == "quflow" or module.startswith("quflow."):
            import importlib
            parts = module.split(".")[1:]
            for depth in range(len(parts), -1, -1):
                try:
                    mod = importlib.import_module(".".join(["quflow_amd"] + parts[:depth]))
                except ImportError:
                    continue
                obj = mod
                try:
                    for attr in name.split("."):
                        obj = getattr(obj, attr)
                except AttributeError:
                    continue
                return obj
        return super().find_class(module, name)


def _loads(data):
    import io
    return _ReferenceUnpickler(io.BytesIO(bytes(data))).load()


# ------------------------------------------------------------------------------------------------
# storage backends
# ------------------------------------------------------------------------------------------------

class DirectoryStore:
    """A directory: `<dataset>.bin` raw rows + `meta.json` (dtype, row shape, row count per dataset;
    attributes of the data path and of args/, bytes values base64-encoded)."""

    def __init__(self, path, create):
        self.path = str(path)
        self.meta_file = os.path.join(self.path, "meta.json")
        if create:
            # Only a quflow record is ever removed (meta.json present; Simulation asks for `create` on an existing
            # record only with overwrite=True).  A directory that is not a record is used if it is empty and
            # refused otherwise; a file of that name is refused: nothing foreign is deleted.
            if os.path.isdir(self.path):
                if os.path.isfile(self.meta_file):
                    shutil.rmtree(self.path)
                elif os.listdir(self.path):
                    raise FileExistsError("'%s' exists, is not empty and is not a quflow_amd record (no meta.json): "
                                          "refusing to replace it" % self.path)
            elif os.path.exists(self.path):
                raise FileExistsError("'%s' exists and is not a directory: refusing to replace it" % self.path)
            os.makedirs(self.path, exist_ok=True)
            self.meta = {"datasets": {}, "attrs": {}, "args": {}}
            self._flush()
        else:
            self._load()

    def _load(self):
        # several objects may be open on one record (a restart re-opens the file while the first object
        # lives on, tests/test_simulation.py:138-143): the file is the truth, as with the HDF5 backend
        with open(self.meta_file) as f:
            self.meta = json.load(f)

    @staticmethod
    def exists(path):
        return os.path.isfile(os.path.join(str(path), "meta.json"))

    def _flush(self):
        tmp = self.meta_file + ".tmp"
        with open(tmp, "w") as f:
            json.dump(self.meta, f)
        os.replace(tmp, self.meta_file)

    def _file(self, name):
        return os.path.join(self.path, name.replace("/", "__") + ".bin")

    def names(self):
        self._load()
        return list(self.meta["datasets"])

    def create(self, name, first_row):
        arr = np.asarray(first_row)          # (0-d stays 0-d: one scalar per row)
        self._load()
        self.meta["datasets"][name] = {"dtype": arr.dtype.str, "shape": list(arr.shape), "rows": 0}
        open(self._file(name), "wb").close()
        self._flush()
        self.append(name, arr)

    def append(self, name, row):
        self._load()
        d = self.meta["datasets"][name]
        arr = np.asarray(row).astype(np.dtype(d["dtype"]), copy=False).reshape(d["shape"])
        with open(self._file(name), "ab") as f:
            f.write(arr.tobytes())
        d["rows"] += 1
        self._flush()

    def info(self, name):
        self._load()
        d = self.meta["datasets"][name]
        return (d["rows"],) + tuple(d["shape"]), np.dtype(d["dtype"])

    def read(self, name, index=None):
        shape, dtype = self.info(name)
        if shape[0] == 0:
            return np.zeros(shape, dtype=dtype)
        mm = np.memmap(self._file(name), dtype=dtype, mode="r", shape=shape)
        out = mm[:] if index is None else mm[index]
        return np.array(out)

    @staticmethod
    def _enc(v):
        if isinstance(v, (bytes, bytearray)):
            return {"__bytes__": base64.b64encode(bytes(v)).decode("ascii")}
        if isinstance(v, np.generic):
            return v.item()
        if isinstance(v, np.ndarray):
            return {"__ndarray__": v.tolist(), "dtype": v.dtype.str}
        return v

    @staticmethod
    def _dec(v):
        if isinstance(v, dict) and "__bytes__" in v:
            return base64.b64decode(v["__bytes__"])
        if isinstance(v, dict) and "__ndarray__" in v:
            return np.array(v["__ndarray__"], dtype=np.dtype(v["dtype"]))
        return v

    def set_attr(self, group, name, value):
        self._load()
        if value is None:
            self.meta[group].pop(name, None)
        else:
            self.meta[group][name] = self._enc(value)
        self._flush()

    def get_attr(self, group, name):
        self._load()
        return self._dec(self.meta[group][name])

    def has_attr(self, group, name):
        self._load()
        return name in self.meta[group]

    def attr_names(self, group):
        self._load()
        return list(self.meta[group])

    def close(self):
        pass


class H5Store:
    """The reference's own HDF5 layout (simulation.py:129-146, 357-431) through h5py."""

    def __init__(self, path, create, datapath="/"):
        import h5py                              # ImportError: the caller reports what is missing
        self.h5py = h5py
        self.path = str(path)
        self.datapath = datapath
        self.args_datapath = datapath + "args/"
        if create:
            with h5py.File(self.path, "w") as f:
                if datapath != "/":
                    f.create_group(datapath)
                f.create_group(self.args_datapath)

    @staticmethod
    def exists(path):
        return os.path.isfile(str(path))

    def _grp(self, f, group):
        return f[self.datapath] if group == "attrs" else f[self.args_datapath]

    def names(self):
        with self.h5py.File(self.path, "r") as f:
            return [n for n in f[self.datapath].keys() if isinstance(f[self.datapath + n], self.h5py.Dataset)]

    def create(self, name, first_row):
        arr = np.asarray(first_row)
        with self.h5py.File(self.path, "r+") as f:
            ds = f.create_dataset(self.datapath + name, (1,) + arr.shape, dtype=arr.dtype, maxshape=(None,) + arr.shape,
                                  chunks=((1,) + arr.shape) if arr.ndim >= 2 else None)
            ds[0, ...] = arr

    def append(self, name, row):
        with self.h5py.File(self.path, "r+") as f:
            ds = f[self.datapath + name]
            ds.resize(ds.shape[0] + 1, axis=0)
            ds[-1, ...] = row

    def info(self, name):
        with self.h5py.File(self.path, "r") as f:
            ds = f[self.datapath + name]
            return tuple(ds.shape), ds.dtype

    def read(self, name, index=None):
        with self.h5py.File(self.path, "r") as f:
            ds = f[self.datapath + name]
            return ds[:] if index is None else ds[index]

    def set_attr(self, group, name, value):
        with self.h5py.File(self.path, "r+") as f:
            g = self._grp(f, group)
            if value is None:
                g.attrs.pop(name)
            elif isinstance(value, (bytes, bytearray)):
                g.attrs[name] = np.array([bytes(value)])        # pickles as in the reference (:139, 216)
            else:
                g.attrs[name] = value

    def get_attr(self, group, name):
        with self.h5py.File(self.path, "r") as f:
            v = self._grp(f, group).attrs[name]
        if isinstance(v, np.ndarray) and v.dtype.kind in "SO" and v.shape == (1,):
            return bytes(v[0])
        return v

    def has_attr(self, group, name):
        with self.h5py.File(self.path, "r") as f:
            return name in self._grp(f, group).attrs

    def attr_names(self, group):
        with self.h5py.File(self.path, "r") as f:
            return list(self._grp(f, group).attrs)

    def close(self):
        pass


def _is_h5_name(filename):
    return str(filename).lower().endswith((".h5", ".hdf5", ".hdf"))


def open_store(filename, create, datapath="/"):
    if _is_h5_name(filename):
        try:
            return H5Store(filename, create, datapath)
        except ImportError as exc:
            raise ImportError("'%s' asks for the HDF5 backend, but h5py is not installed; use a directory name "
                              "for the dependency-free backend" % filename) from exc
    return DirectoryStore(filename, create)


def store_exists(filename):
    return H5Store.exists(filename) if _is_h5_name(filename) else DirectoryStore.exists(filename)


# ------------------------------------------------------------------------------------------------
# the simulation record
# ------------------------------------------------------------------------------------------------

class Simulation:
    """Counterpart of quflow.QuSimulation (simulation.py:60-478): output record, checkpoint and
    callback of `solve` in one object.  See the module docstring for the layout."""

    def __init__(self, filename, overwrite=False, state=None, time=None, qutypes=None, loggers=None, datapath="/",
                 **kwargs):
        from . import __version__
        self.filename = str(filename)
        if datapath[-1] != "/":
            raise ValueError("Datapath must end with /")
        self.datapath = datapath
        self.loggers = loggers if loggers is not None else dict()
        if not store_exists(filename) or overwrite:
            if state is None:
                raise ValueError("At least `state` must be provided to initialize a Simulation.")
            self.qutypes = dict(_default_qutypes if qutypes is None else qutypes)
            for q in self.qutypes:
                if q not in _qutype2varname:
                    raise NotImplementedError("qutype '%s' (function-space output, quflow/transforms.py) is outside "
                                              "this package; available: %s" % (q, sorted(_qutype2varname)))
            self.store = open_store(filename, True, datapath)
            self.store.set_attr("attrs", "version", __version__)
            self.store.set_attr("attrs", "created", datetime.datetime.now().isoformat())
            self.store.set_attr("attrs", "qutypes", pickle.dumps(self.qutypes))
            try:
                self.store.set_attr("attrs", "loggers", pickle.dumps(self.loggers))
            except (AttributeError, pickle.PicklingError, TypeError):
                pass                                   # (local functions do not pickle: as in the reference, :140-145)
            self._initialize_fields(np.asarray(state), 0.0 if time is None else time, kwargs)
        else:
            self.store = open_store(filename, False, datapath)
            if state is not None and self.store.has_attr("attrs", "N"):
                raise ValueError(self.filename + " has already been initialized with W.")
            if qutypes is not None:
                raise ValueError(self.filename + " has already been initialized with qutypes.")
            self.qutypes = _loads(self.store.get_attr("attrs", "qutypes"))
            if self.store.has_attr("attrs", "loggers") and loggers is None:
                self.loggers = _loads(self.store.get_attr("attrs", "loggers"))

    # ---- representations of the state (simulation.py:285-343, the matrix and coefficient forms)
    def _representations(self, W, device_shr=None):
        N = W.shape[-1]
        for qutype, dtype in self.qutypes.items():
            if qutype == 'mat':
                arr = W.astype(W.dtype if dtype is None else dtype)
            elif qutype == 'shr':
                from .quantization import mat2shr
                rows = device_shr if device_shr is not None else [mat2shr(Wi) for Wi in W.reshape((-1, N, N))]
                arr = np.squeeze(np.array(rows))
                arr = arr.astype(W.real.dtype if dtype is None else dtype)
            elif qutype != 'shc':
                # (a file the reference wrote with its default qutypes holds 'fun' / 'funL2' rows: reading them works,
                # appending would need the spherical-harmonics synthesis of quflow/transforms.py)
                raise NotImplementedError("qutype '%s' (function-space output, quflow/transforms.py) is outside this "
                                          "package: '%s' can be read but not appended to" % (qutype, self.filename))
            else:
                from .quantization import mat2shc
                arr = np.squeeze(np.array([mat2shc(Wi) for Wi in W.reshape((-1, N, N))]))
                arr = arr.astype(W.dtype if dtype is None else dtype)
            yield _qutype2varname[qutype], arr, qutype

    def _initialize_fields(self, W, time, fields):
        for varname, arr, qutype in self._representations(W):
            self.store.create(varname, arr)
        self.store.set_attr("attrs", "N", int(W.shape[-1]))
        self.store.create("time", np.float64(time))
        self.store.create("step", np.int64(0))
        for name, logger in self.loggers.items():
            self.store.create(name, np.asarray(logger(W)))
        fields = dict(fields)
        for name in _stats_fields:                     # simulation.py:409-412
            fields.setdefault(name, 0.0)
        for name, value in fields.items():
            if name in ("time", "step"):
                raise ValueError("{} is not a valid field name.".format(name))
            self.store.create(name, np.asarray(value))

    # ---- the callback protocol of solve (simulation.py:433-478)
    def __call__(self, W, delta_time, delta_steps=1, device_shr=None, **kwargs):
        W = np.asarray(W)
        rows = list(self._representations(W, device_shr))      # every row first: a refused qutype appends nothing
        for varname, arr, qutype in rows:
            self.store.append(varname, arr)
        self.store.append("time", self.store.read("time", -1) + delta_time)
        self.store.append("step", self.store.read("step", -1) + delta_steps)
        names = set(self.store.names())
        for varname, value in kwargs.items():
            if varname in names and varname not in self.loggers:
                self.store.append(varname, value)
        for name, logger in self.loggers.items():
            self.store.append(name, logger(W))

    # ---- item access (simulation.py:203-276)
    def __setitem__(self, name, value):
        if name in _pickled_argnames:
            if value is None:
                self.store.set_attr("args", name, None)
                return
            try:
                self.store.set_attr("args", name, pickle.dumps(value))
            except (AttributeError, pickle.PicklingError, TypeError):
                self.store.set_attr("args", name, value.__name__)
        elif name in _info_args or name == "prerun":
            self.store.set_attr("attrs", name, value)
        else:
            self.store.set_attr("args", name, value)

    def __getitem__(self, name):
        ind = None
        if isinstance(name, tuple) and isinstance(name[0], str):
            ind = name[1:] if len(name) > 2 else name[1]
            name = name[0]
        if not isinstance(name, str):
            ind, name = name, "mat"                    # an index alone means the state
        if name in self.store.names():
            return self.store.read(name, ind)
        if self.store.has_attr("args", name):
            v = self.store.get_attr("args", name)
            if name in _pickled_argnames:
                return _loads(v) if isinstance(v, (bytes, bytearray)) else v
            return v
        if self.store.has_attr("attrs", name):
            v = self.store.get_attr("attrs", name)
            return _loads(v) if name == "qutypes" else v
        raise KeyError("There is no dataset or attribute '{}'.".format(name))

    def args(self):
        for name in self.store.attr_names("args"):
            yield name, self[name]

    @property
    def fieldnames(self):
        return {name: self.store.info(name) for name in self.store.names()}


# ------------------------------------------------------------------------------------------------
# the chunk driver
# ------------------------------------------------------------------------------------------------

# ------------------------------------------------------------------------------------------------
# run file
# ------------------------------------------------------------------------------------------------

_RUNFILE = '''#!/usr/bin/env python3
# Run file of the simulation record {record!r}, written by quflow_amd.create_runfile.
# Continues the record with the MI355X stepper: the trajectory stays resident on the device between output chunks.
import argparse
import os
import sys

import numpy as np
import quflow_amd as qf

parser = argparse.ArgumentParser()
parser.add_argument("-f", "--filename", help="simulation record (HDF5 file or directory)", type=str,
                    default=os.path.join(os.path.dirname(os.path.abspath(__file__)), {basename!r}))
parser.add_argument("-t", "--simtime", help="total simulation time of this run", type=float)
parser.add_argument("--steps", help="number of steps of this run", type=int)
parser.add_argument("--compsum", help="compensated summation", action="store_true")
parser.add_argument("--tol", help="tolerance of the fixed-point iteration", type=float)
args = parser.parse_args()

# ---------- externally defined code (the record's `prerun`) ----------
{prerun}
# ----------------------------------------------------------------------

if qf.device_count() < 1:
    sys.exit("no HIP device visible: quflow_amd has no CPU path")
mysim = qf.QuSimulation(args.filename)
solve_kwargs = dict()
if args.simtime is not None:
    solve_kwargs["simtime"] = np.float64(args.simtime)
if args.steps is not None:
    solve_kwargs["steps"] = int(args.steps)
if args.tol is not None:
    solve_kwargs["tol"] = np.float64(args.tol)
if args.compsum:
    solve_kwargs["compsum"] = True
W = qf.solve(mysim, progress_bar=False, **solve_kwargs)
print("%s: step %d, time %.6g, rows %d" % (args.filename, int(mysim["step", -1]), float(mysim["time", -1]), mysim.fieldnames["mat"][0][0]))
'''


def create_runfile(sim, runfilename=None):
    """Counterpart of quflow.simulation.create_runfile (simulation.py:484-585): writes a stand-alone script next to the
    record that re-opens it and continues the run with the stored arguments -- the reference's generated script picks
    device objects when `hamiltonian is qf.solve_poisson and integrator is qf.isomp` (:554-562); here the stored pair IS the
    device pair (pickled `quflow.*` callables map to this package at unpickling), so the script just calls `solve`.
    The record's `prerun` code (user definitions the pickled callables need) is pasted in as the reference does.  No
    animation step: graphics are outside this package.  Returns the path of the script."""
    if not isinstance(sim, Simulation):
        sim = Simulation(str(sim))
    try:
        prerun = sim['prerun']
    except KeyError:
        prerun = ""
    if runfilename is None:
        base = sim.filename
        for ext in (".hdf5", ".h5", ".hdf", ".qf"):
            if base.lower().endswith(ext):
                base = base[:-len(ext)]
                break
        runfilename = base + "_runfile.py"
    text = _RUNFILE.format(record=os.path.basename(sim.filename), basename=os.path.basename(sim.filename), prerun=str(prerun).strip())
    with open(runfilename, "w") as f:
        f.write(text)
    return runfilename


def _device_resident_ok(integrator, kwargs):
    """The default stepper without host hooks: the trajectory may stay on the device between chunks."""
    from . import integrators as _int
    from . import laplacian as _lap
    if integrator not in (_int.isomp, _int.isomp_fixedpoint) and not isinstance(integrator, _int.IsompHIP):
        return False
    if any(kwargs.get(k) is not None for k in ("forcing", "strang_splitting", "callback")):
        return False
    if any(k not in ("time", "hamiltonian", "stats", "tol", "maxit", "minit", "compsum", "reinitialize", "verbatim")
           for k in kwargs):
        return False
    return _int._is_native_hamiltonian(kwargs.get("hamiltonian")) and _int._SKEW_HERM_ and _lap._SKEW_HERM_


def _in_notebook():
    try:
        from IPython import get_ipython
        return 'IPKernelApp' in get_ipython().config
    except Exception:
        return False


def solve(W, dt=None, stepsize=None, steps=None, simtime=None, endtime=None, steps_out=None, dt_out=None,
          integrator=None, callback=None, callback_kwargs=None, integrator_callback=None,
          progress_bar=True, progress_file=None, inner_steps=None, inner_time=None, resident=None, **kwargs):
    """The chunk loop of quflow.simulation.solve (simulation.py:604-798): `W` is a state matrix or a
    `Simulation` to continue (its last row, its time, its stored arguments).  Every `steps_out` steps
    the callbacks -- a `Simulation` among them -- get `(W, delta_time=..., delta_steps=..., **stats)`.
    progress_bar / progress_file: the reference's tqdm progress display (simulation.py:764-780); never
    forwarded to the integrator.  inner_steps / inner_time: the deprecated names of steps_out / dt_out.
    resident: keep the trajectory on the device between the chunks (default: whenever the stepper is
    quflow_amd.isomp with its built-in Hamiltonian and no host hooks, on a complex128 (N,N) state).
    As in the reference the caller's array is advanced in place (isomp overwrites W) and the final
    state is returned."""
    from . import integrators as _int
    from . import laplacian as _lap
    time = kwargs.get("time", 0.0)
    if steps_out is None:
        steps_out = inner_steps
    if dt_out is None:
        dt_out = inner_time
    if isinstance(W, Simulation):
        sim = W
        W = sim['mat', -1]
        time = float(sim['time', -1])
        callback = sim if callback is None else (tuple(callback) if isinstance(callback, tuple) else (callback,)) + (sim,)
        stored = dict(sim.args())
        dt = stored.get('dt') if dt is None else dt
        stepsize = stored.get('stepsize') if stepsize is None else stepsize
        if steps is None and simtime is None and endtime is None:
            steps, simtime, endtime = stored.get('steps'), stored.get('simtime'), stored.get('endtime')
        if steps_out is None and dt_out is None:
            steps_out = stored.get('steps_out', stored.get('inner_steps'))
            dt_out = stored.get('dt_out', stored.get('inner_time'))
        integrator = stored.get('integrator') if integrator is None else integrator
        if integrator_callback is None:
            integrator_callback = stored.get('integrator_callback', stored.get('callback'))
        callback_kwargs = stored.get('callback_kwargs') if callback_kwargs is None else callback_kwargs
        if progress_bar is None:
            progress_bar = stored.get('progress_bar')
        if progress_file is None:
            progress_file = stored.get('progress_file')
        for name, value in stored.items():
            if name not in ('dt', 'stepsize', 'steps', 'simtime', 'endtime', 'steps_out', 'inner_steps', 'dt_out', 'inner_time',
                            'integrator', 'integrator_callback', 'callback', 'callback_kwargs', 'progress_bar', 'progress_file'):
                kwargs.setdefault(name, value)
    W = np.asarray(W)
    N = W.shape[-1]
    if dt is None:
        if stepsize is None:
            raise ValueError("Either `dt` or `stepsize` must be specified.")
        dt = stepsize * hbar(N)
    if integrator is None:
        integrator = _int.isomp
    ikw = kwargs
    ikw['time'] = time
    ikw.setdefault('hamiltonian', _lap.solve_poisson)
    if 'stats' in inspect.getfullargspec(integrator).args:      # (works for callable objects too, simulation.py:730)
        ikw['stats'] = {'iterations': 0.0}
    if integrator_callback is not None:
        ikw['callback'] = integrator_callback
    if sum(x is not None for x in (steps, simtime, endtime)) != 1:
        warnings.warn("One, and only one, of `steps`, `simtime`, or `endtime` should be specified.")
    if endtime is not None:
        if endtime < time:
            raise ValueError("Specified `endtime`={} is smaller than current `time`={}.".format(endtime, time))
        simtime = endtime - time
    if simtime is not None:
        steps = round(simtime / np.abs(dt))
    if callback is not None and not isinstance(callback, tuple):
        callback = (callback,)
    callback_kwargs = dict() if callback_kwargs is None else callback_kwargs
    if steps_out is None:
        steps_out = 100 if dt_out is None else round(dt_out / np.abs(dt))
    steps_out = min(steps_out, steps)

    pbar = None
    if progress_bar:                                   # simulation.py:764-780
        try:
            if progress_file is None:
                if not ikw.get('verbatim'):
                    if _in_notebook():
                        from tqdm.notebook import tqdm
                    else:
                        from tqdm import tqdm
                    pbar = tqdm(total=steps, unit=' steps')
            else:
                from tqdm import tqdm
                pbar = tqdm(total=steps, unit=' steps', file=progress_file, ascii=True, mininterval=10.0)
        except ModuleNotFoundError:
            pbar = None

    W_caller = W
    use_device = _device_resident_ok(integrator, ikw) if resident is None else bool(resident)
    tr = None
    # (a complex64 state makes a single-precision resident trajectory: float32 solve, complex64 products and the
    # float32 tolerance rule, exactly what isomp does with a complex64 host array)
    if use_device and W.ndim == 2 and W.dtype in (np.complex128, np.complex64):
        tr = _int.DeviceTrajectory(W)
        adv_kw = {k: ikw[k] for k in ("tol", "maxit", "minit", "compsum", "reinitialize") if k in ikw}
    want_shr = any(isinstance(c, Simulation) and 'shr' in c.qutypes for c in (callback or ()))
    try:
        for k0 in range(0, steps, max(steps_out, 1)):
            n = min(steps_out, steps - k0)
            extra = {}
            if tr is not None:
                st = tr.advance(dt, n, **adv_kw)
                W = tr.download()
                if W_caller.flags.writeable:
                    W_caller[...] = W                  # the reference's stepper advances the caller's array in place
                if 'stats' in ikw and ikw['stats']:
                    if isinstance(adv_kw.get("tol", 'auto'), str) or adv_kw.get("tol", -1) < 0:
                        ikw['stats']['tol_auto'] = st["tol"]
                    ikw['stats']['iterations'] = st["iterations"]
                    ikw['stats']['number_of_maxit'] = st["number_of_maxit"]
                if want_shr and not tr.c64:
                    extra["device_shr"] = [tr.shr()]        # mat2shr on the resident state: N^2 doubles over PCIe
            else:
                W = integrator(W, dt, steps=n, **ikw)
            delta_time = n * dt
            ikw['time'] += delta_time
            if pbar is not None:
                pbar.update(n)
            for cfun in (callback or ()):
                if 'stats' in ikw:
                    callback_kwargs.update(ikw['stats'])
                if isinstance(cfun, Simulation):
                    cfun(W, delta_time=delta_time, delta_steps=n, **extra, **callback_kwargs)
                else:
                    cfun(W, delta_time=delta_time, delta_steps=n, **callback_kwargs)
    finally:
        if tr is not None:
            tr.ctx.close()
        if pbar is not None:
            pbar.close()
    return W
