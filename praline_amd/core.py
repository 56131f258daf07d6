"""Minimal host runtime that carries the hot-path components.

Only the surface the reference's callers use to reach the pairwise aligner is restated here
(reference: praline/core/component.py, manager.py, execution.py, exception.py):
  * Component / Container / Port / T / Environment and the Message types,
  * TypeIndex.register / resolve (praline/core/manager.py:33-99),
  * Manager.execute_one / execute_many / _invoke with port and option checking
    (manager.py:134-232) - execute_many is the seam where BatchManager (component.py) turns a
    homogeneous PairwiseAligner request list into ONE device submission,
  * Execution / Task (praline/core/execution.py:119-290).
Parallel / remote managers, the TaskNode progress tree and LogBundle are orchestration and out of
scope.
"""
import warnings
import uuid

MESSAGE_KIND_BEGIN = "begin"
MESSAGE_KIND_PROGRESS = "progress"
MESSAGE_KIND_COMPLETE = "complete"
MESSAGE_KIND_ERROR = "error"
MESSAGE_KIND_LOG = "log"

ENTRY_POINT_GROUP = "praline.type"   # praline/core/component.py: the setuptools group TypeIndex.autoregister reads

ROOT_TAG = "__ROOT_TAG__"

_PRIMITIVES = (int, float, str, bool)


# ---- exceptions (praline/core/exception.py) ----------------------------------------------------
class PralineError(Exception):
    pass


class AlphabetError(PralineError):
    pass


class SequenceError(PralineError):
    pass


class SignatureError(PralineError):
    pass


class ComponentError(PralineError):
    pass


class MessageError(PralineError):
    pass


class DataError(PralineError):
    pass


# ---- type signatures -----------------------------------------------------------------------------
class T(object):
    """Nullable wrapper for a type in a signature (praline/core/component.py:328-340)."""

    def __init__(self, tid, nullable=False):
        self.tid = tid
        self.nullable = nullable


class Port(object):
    """Input / output port: a type signature and an optional flag (component.py:204-216)."""

    def __init__(self, signature, optional=False):
        self.signature = signature
        self.optional = optional


def check_signature(sig):
    """Well-formedness of a signature (component.py:342-372): [one item], non-empty tuples, type
    id strings, T(...) or a primitive type."""
    if isinstance(sig, list):
        if len(sig) != 1:
            raise SignatureError("lists in signatures must contain exactly one item")
        check_signature(sig[0])
    elif isinstance(sig, tuple):
        if not sig:
            raise SignatureError("tuples in signatures must be non-empty")
        for item in sig:
            check_signature(item)
    elif isinstance(sig, str):
        return
    elif isinstance(sig, T):
        check_signature(sig.tid)
    elif sig in _PRIMITIVES:
        return
    else:
        raise SignatureError("type invalid for signature: {0}".format(type(sig)))


def conforms(sig, value):
    """Does value conform to signature?  Raises DataError otherwise (manager.py:566-615)."""
    if isinstance(sig, T):
        if value is None:
            if not sig.nullable:
                raise DataError("value may not be None")
            return
        conforms(sig.tid, value)
    elif isinstance(sig, list):
        if not isinstance(value, (list, tuple)) and not hasattr(value, "__iter__"):
            raise DataError("expected a list, got {0}".format(type(value)))
        for item in value:
            conforms(sig[0], item)
    elif isinstance(sig, tuple):
        if len(value) != len(sig):
            raise DataError("expected a {0}-tuple".format(len(sig)))
        for s, v in zip(sig, value):
            conforms(s, v)
    elif isinstance(sig, str):
        tid = getattr(value, "tid", None)
        if tid is None:
            raise DataError("expected an object of type '{0}', got {1}".format(sig, type(value)))
        # subclasses conform to their parents' type ids
        tids = [getattr(c, "tid", None) for c in type(value).__mro__]
        if sig not in tids:
            raise DataError("expected type id '{0}', got '{1}'".format(sig, tid))
    elif sig is float:
        if isinstance(value, bool) or not isinstance(value, (int, float)) and not hasattr(value, "__float__"):
            raise DataError("expected a float, got {0}".format(type(value)))
    elif sig is int:
        if isinstance(value, bool) or not (isinstance(value, int) or hasattr(value, "__index__")):
            raise DataError("expected an int, got {0}".format(type(value)))
    elif sig in (str, bool):
        if not isinstance(value, sig):
            raise DataError("expected {0}, got {1}".format(sig.__name__, type(value)))


# ---- containers / components ---------------------------------------------------------------------
class Container(object):
    """Base of every non-primitive datum passed between components (component.py:60-97)."""
    tid = "praline.container.Container"


class Environment(Container):
    """Key/value options with three-level inheritance: component defaults <- parent environment
    <- explicit keys, recursive for nested environments (component.py:99-201)."""
    tid = "praline.container.Environment"

    def __init__(self, keys=None, component=None, parent=None):
        sources = []
        if component:
            sources.append(component.defaults)
        if parent:
            sources.append(parent.keys)
        if keys:
            sources.append(keys)
        self.keys = self._inherit(sources)

    @staticmethod
    def _inherit(sources):
        merged = {}
        for source in sources:
            merged.update(source)
        for key, value in list(merged.items()):
            if isinstance(value, Environment):
                nested = [src[key].keys for src in sources
                          if key in src and isinstance(src[key], Environment)]
                merged[key] = Environment(Environment._inherit(nested))
        return merged

    def collapse(self, component, env):
        return Environment(keys=env.keys, component=component, parent=self)

    def __getitem__(self, key):
        return self.keys[key]

    def get(self, key, default=None):
        return self.keys.get(key, default)


class Component(object):
    """Component superclass: class attributes tid / inputs / outputs / options / defaults and a
    generator execute(**inputs) that ends with a CompleteMessage (component.py:22-57)."""
    tid = "praline.component.Component"
    inputs = {}
    outputs = {}
    options = {}
    defaults = {}

    def __init__(self, manager, environment, tag):
        for port in list(self.inputs.values()) + list(self.outputs.values()):
            check_signature(port.signature)
        for sig in self.options.values():
            check_signature(sig)
        self.manager = manager
        self.environment = environment
        self.tag = tag

    def execute(self, **kwargs):
        raise NotImplementedError("please override execute() in your Component subclass")


# ---- messages (component.py:219-326) -------------------------------------------------------------
class Message(object):
    def __init__(self, kind):
        self.kind = kind
        self.tag = None


class BeginMessage(Message):
    def __init__(self, parent_tag=None):
        Message.__init__(self, MESSAGE_KIND_BEGIN)
        self.parent_tag = parent_tag


class ProgressMessage(Message):
    def __init__(self, progress):
        Message.__init__(self, MESSAGE_KIND_PROGRESS)
        if progress > 1.0 or progress < 0.0:
            raise MessageError("progress should be a float value between 0.0 and 1.0")
        self.progress = progress


class CompleteMessage(Message):
    def __init__(self, outputs):
        Message.__init__(self, MESSAGE_KIND_COMPLETE)
        self.outputs = outputs


class ErrorMessage(Message):
    def __init__(self, error):
        Message.__init__(self, MESSAGE_KIND_ERROR)
        self.error = error


# ---- type index + manager ------------------------------------------------------------------------
class TypeIndex(object):
    """tid string -> class registry (manager.py:33-99).  autoregister() registers the components
    of this package (the reference reads the 'praline.type' entry-point group, setup.py:8-20)."""

    def __init__(self):
        self._types = {}

    def register(self, component_class):
        self._types[component_class.tid] = component_class

    def unregister(self, component_class):
        if component_class.tid not in self._types:
            raise ComponentError("component with type id '{0}' not registered".format(component_class.tid))
        del self._types[component_class.tid]

    def autoregister(self, strict=False):
        """Register every component published under the `praline.type` entry-point group - what the reference does
        (manager.py:72-85): this package's own components (setup.py) and any third-party aligner installed beside it.
        Without packaging metadata (a bare source tree) the in-package list is registered."""
        # This package's components come first and are never displaced: the group is shared with the reference
        # distribution (`praline-aln` publishes the same tids), and with both installed "last one wins" would hand the
        # managers of this package foreign Component classes in unspecified order.
        from . import component
        own = {}
        for cls in component.COMPONENTS:
            self.register(cls)
            own[cls.tid] = cls
        try:
            from importlib import metadata
            eps = metadata.entry_points()
            group = eps.select(group=ENTRY_POINT_GROUP) if hasattr(eps, "select") else eps.get(ENTRY_POINT_GROUP, [])
        except Exception as exc:   # no packaging metadata machinery: the in-package list stands
            warnings.warn("praline_amd: entry-point discovery failed (%s): only the package's own components are "
                          "registered" % (exc,), RuntimeWarning, stacklevel=2)
            return
        # Every entry point that is NOT taken is reported (the reference, manager.py:72-85, propagates load errors and
        # lets the last registration win; here the package's own tids are never displaced - INTEGRATION.md section 4 - so
        # a user whose replacement aligner is ignored must be told).  `strict=True` restores the reference's behaviour
        # for load errors: they propagate.
        for entry_point in group:
            try:
                cls = entry_point.load()
            except Exception as exc:
                if strict:
                    raise
                # a distribution whose import fails here (the reference itself without its dependencies) is skipped
                warnings.warn("praline_amd: entry point %r of group %r could not be loaded (%s: %s) - skipped"
                              % (getattr(entry_point, "name", entry_point), ENTRY_POINT_GROUP, type(exc).__name__, exc),
                              RuntimeWarning, stacklevel=2)
                continue
            # third-party components must be written against THIS runtime (a subclass of its Component) and may
            # add tids, not replace the package's own
            if not (isinstance(cls, type) and issubclass(cls, Component)):
                if getattr(cls, "__module__", "").split(".")[0] != "praline":   # (the reference's own classes: expected)
                    warnings.warn("praline_amd: entry point %r is not a praline_amd.core.Component subclass - skipped"
                                  % (getattr(entry_point, "name", entry_point),), RuntimeWarning, stacklevel=2)
                continue
            tid = getattr(cls, "tid", None)
            if tid is None:
                continue
            if tid in own and own[tid] is not cls:
                warnings.warn("praline_amd: entry point %r publishes type id %r, which belongs to this package: its own "
                              "component is kept (register the replacement explicitly with TypeIndex.register to override)"
                              % (getattr(entry_point, "name", entry_point), tid), RuntimeWarning, stacklevel=2)
                continue
            self.register(cls)

    def resolve(self, tid):
        try:
            return self._types[tid]
        except KeyError:
            raise ComponentError("component with type id '{0}' not registered".format(tid))


class Manager(object):
    """Serial manager: instantiates components, checks options / ports, runs execute()
    (manager.py:117-240)."""

    def __init__(self, index):
        self.index = index
        self.open = True

    def _require_open(self):
        if not self.open:
            raise PralineError("manager has been closed")

    def execute_one(self, request, parent_tag):
        self._require_open()
        tid, inputs, tag, env = request
        for message in self._invoke(tid, inputs, tag, env, parent_tag=parent_tag):
            yield message

    def execute_many(self, requests, parent_tag):
        self._require_open()
        for tid, inputs, tag, env in requests:
            for message in self._invoke(tid, inputs, tag, env, parent_tag=parent_tag):
                yield message

    def _check_request(self, component, inputs, environment):
        for key, sig in component.options.items():
            conforms(sig, environment[key])
        for name, port in component.inputs.items():
            inputs[name] = inputs.get(name, None)
            if inputs[name] is None:
                if not port.optional:
                    raise DataError("input '{0}' is not optional but was not supplied".format(name))
            else:
                conforms(port.signature, inputs[name])

    @staticmethod
    def _check_outputs(component, outputs):
        for name, port in component.outputs.items():
            outputs[name] = outputs.get(name, None)
            if outputs[name] is None:
                if not port.optional:
                    raise DataError("output '{0}' is not optional but was not supplied".format(name))
            else:
                conforms(port.signature, outputs[name])

    def _invoke(self, tid, inputs, tag, environment, submanager=None, parent_tag=None):
        component = self.index.resolve(tid)(submanager or self, environment, tag)
        self._check_request(component, inputs, environment)
        begin = BeginMessage(parent_tag)
        begin.tag = tag
        yield begin
        for message in component.execute(**inputs):
            if not isinstance(message, Message):
                raise TypeError("component messages should be subclasses of Message")
            if message.tag is None:
                message.tag = tag
            if message.kind == MESSAGE_KIND_COMPLETE and message.tag == tag:
                self._check_outputs(component, message.outputs)
            yield message

    def close(self):
        self._require_open()
        self.open = False


# ---- execution helper (praline/core/execution.py:119-290) ----------------------------------------
def _generate_tag(tid):
    return "{0}#{1}".format(tid.split(".")[-1], uuid.uuid4().hex)


class Task(object):
    def __init__(self, execution, component, tag):
        self.tag = tag
        self._component = component
        self._inputs = None
        self._env = None
        self._root_env = None

    def inputs(self, **kwargs):
        if self._inputs is None:
            self._inputs = {}
        self._inputs.update(kwargs)
        return self

    def environment(self, root_env=None, env=None):
        self._root_env = root_env
        self._env = env
        return self

    def get(self):
        if self._inputs is None:
            raise ComponentError("please provide inputs for this execution task")
        env = self._env if self._env is not None else Environment({})
        root_env = self._root_env if self._root_env is not None else Environment({})
        return self._component, root_env.collapse(self._component, env), self._inputs, self.tag


class Execution(object):
    def __init__(self, manager, parent_tag=None, strip_bulk_data=True):
        self.manager = manager
        self.parent_tag = parent_tag
        self.strip_bulk_data = strip_bulk_data
        self._tasks = []
        self._tags = set()
        self._outputs = None
        self._done = False

    def add_task(self, component):
        tag = _generate_tag(component.tid)
        self._tags.add(tag)
        task = Task(self, component, tag)
        self._tasks.append(task)
        return task

    def run(self):
        tag_index = {}
        requests = []
        self._outputs = [None] * len(self._tasks)
        for i, task in enumerate(self._tasks):
            component, env, inputs, tag = task.get()
            tag_index[tag] = i
            requests.append((component.tid, inputs, tag, env))
        for message in self.manager.execute_many(requests, self.parent_tag):
            if message.kind == MESSAGE_KIND_COMPLETE:
                if message.tag in tag_index:
                    self._outputs[tag_index[message.tag]] = message.outputs
                if self.strip_bulk_data:
                    message.outputs = None
            yield message
        self._done = True

    def started_task(self, tag):
        return tag in self._tags

    @property
    def outputs(self):
        if not self._done:
            raise ComponentError("cannot access outputs until all messages have been consumed "
                                 "from Execution.run()")
        return self._outputs


def run(execution):
    """Drain an execution and return its outputs list."""
    for _ in execution.run():
        pass
    return execution.outputs
