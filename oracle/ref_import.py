"""
TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Loads the upstream Python reference from /root/reference so that golden vectors
can be generated from it (oracle/gen_golden.py).  Runs only in the build container;
the reference never travels to the GPU box.

Two adaptations are needed to import the reference on this image, neither touches
arithmetic of the simulate/attract/target path:

* the storage packages ZODB / BTrees / persistent / transaction are not installed;
  attract.py, simulate.py, target.py and mpi.py import them at module top for their
  result *storage* only.  Minimal in-memory stand-ins are registered in sys.modules
  (dict-backed BTree with sorted items(), integer Length cell, no-op transaction).
* input.py:884 calls exec(..., locals=...), a keyword only Python >= 3.13 accepts;
  on this image's 3.10 every YAML with update rules raises TypeError.  The single
  function build_truth_table_from_safe_update_rule is replaced by an equivalent that
  evaluates the same safe rule text with eval() over the same truth-table rows.
"""
import os
import sys
import types
import itertools

REFERENCE_ROOT = '/root/reference'


class _BTree(dict):
    def items(self):
        return iter(sorted(dict.items(self)))

    def keys(self):
        return iter(sorted(dict.keys(self)))

    def values(self):
        return iter(v for _, v in sorted(dict.items(self)))


class _Length:
    def __init__(self, value=0):
        self.value = value

    def __call__(self):
        return self.value

    def change(self, delta):
        self.value += delta

    def set(self, value):
        self.value = value


class _Root:
    pass


class _DB:
    def pack(self):
        pass


class _Conn:
    def __init__(self):
        self.root = _Root()
        self._db = _DB()

    def db(self):
        return self._db

    def cacheMinimize(self):
        pass

    def close(self):
        pass


def _install_storage_stand_ins():
    persistent = types.ModuleType('persistent')
    persistent.Persistent = object
    transaction = types.ModuleType('transaction')
    transaction.commit = lambda: None
    transaction.abort = lambda: None
    btrees = types.ModuleType('BTrees')
    oobtree = types.ModuleType('BTrees.OOBTree')
    oobtree.BTree = _BTree
    length = types.ModuleType('BTrees.Length')
    length.Length = _Length
    btrees.OOBTree = oobtree
    btrees.Length = length
    zodb = types.ModuleType('ZODB')
    zodb.connection = lambda *_a, **_k: _Conn()
    zodb.DB = None
    for name, mod in (('persistent', persistent), ('transaction', transaction),
                      ('BTrees', btrees), ('BTrees.OOBTree', oobtree),
                      ('BTrees.Length', length), ('ZODB', zodb)):
        sys.modules.setdefault(name, mod)


def load_reference():
    """Return the imported `boolsi` package of the reference (dict of modules)."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, 'boolsi')):
        raise RuntimeError('reference not present at {} -- golden vectors can only be '
                           'generated in the build container'.format(REFERENCE_ROOT))
    _install_storage_stand_ins()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import boolsi.model as model
    import boolsi.batching as batching
    import boolsi.constants as constants
    import boolsi.input as rinput
    import boolsi.mpi as mpi
    import boolsi.attract as attract
    import boolsi.simulate as simulate
    import boolsi.target as target

    if sys.version_info < (3, 13):
        majority = model.majority

        def build_truth_table(safe_update_rule, predecessor_nodes, safe_node_names, err_msg):
            table = dict()
            for states in itertools.product((False, True), repeat=len(predecessor_nodes)):
                env = {safe_node_names[p]: states[i] for i, p in enumerate(predecessor_nodes)}
                try:
                    table[states] = eval(safe_update_rule, {'majority': majority}, env)
                except BaseException:
                    raise ValueError(err_msg)
                if table[states] not in {False, True}:
                    raise ValueError(err_msg)
            return table

        rinput.build_truth_table_from_safe_update_rule = build_truth_table

    return dict(model=model, batching=batching, constants=constants, input=rinput, mpi=mpi,
                attract=attract, simulate=simulate, target=target, zodb=sys.modules['ZODB'])
