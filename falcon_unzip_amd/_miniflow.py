"""Minimal in-process stand-in for the slice of `pypeflow.simple_pwatcher_bridge` that the phasing
task chain uses (reference call sites: falcon_unzip/phasing.py:1-2,496-553).  Used only when the
real pypeflow is not importable; with pypeflow installed the task functions in
falcon_unzip_amd.phasing are ordinary pypeflow task bodies.

Semantics kept: tasks run in insertion order on refreshTargets(); every output's directory exists
before the task body runs; inputs/outputs are reachable as attributes and through .inputs/.outputs;
`fn()` maps a file handle to its path; a task whose outputs are all newer than its inputs is skipped
only when `skip_up_to_date=True` is asked for (pypeflow's make-style resume).
"""
from __future__ import annotations

import os


class _PypeLocalFile(str):
    pass


def makePypeLocalFile(path):
    return _PypeLocalFile(path)


def fn(f):
    return str(f)


class MyFakePypeThreadTaskBase(object):
    pass


class _Task(object):
    def __init__(self, inputs, outputs, parameters):
        self.inputs = dict(inputs or {})
        self.outputs = dict(outputs or {})
        self.parameters = dict(parameters or {})
        for k, v in list(self.inputs.items()) + list(self.outputs.items()):
            setattr(self, k, v)
        self._func = None
        self.generated_script_fn = None


def PypeTask(inputs=None, outputs=None, parameters=None, **kw):
    def deco(func):
        t = _Task(inputs, outputs, parameters)
        t._func = func
        return t
    return deco


class PypeProcWatcherWorkflow(object):
    def __init__(self, max_jobs=1, skip_up_to_date=False, **kw):
        self.max_jobs = max_jobs
        self._tasks = []
        self._skip = skip_up_to_date

    def addTask(self, t):
        self._tasks.append(t)

    def addTasks(self, ts):
        self._tasks.extend(ts)

    def _up_to_date(self, t):
        try:
            newest_in = max([os.path.getmtime(fn(p)) for p in t.inputs.values()] or [0])
            oldest_out = min(os.path.getmtime(fn(p)) for p in t.outputs.values())
        except OSError:
            return False
        return oldest_out >= newest_in

    def refreshTargets(self, *a, **k):
        tasks, self._tasks = self._tasks, []
        for t in tasks:
            if self._skip and t.outputs and self._up_to_date(t):
                continue
            for p in t.outputs.values():
                d = os.path.dirname(fn(p))
                if d and not os.path.isdir(d):
                    os.makedirs(d)
            t._func(t)
