"""
YAML front end: reads a BoolSi input file and produces (a) the same configuration dict the
reference's `process_input` returns and (b) the flat, bit-packed tables the MI355X engine
consumes (`compile_problem`).

Behavioural contract = reference `boolsi/input.py` (v1.0.5):
  sections and their validation ............ input.py:98-159
  node names ............................... input.py:162-232
  update rules -> predecessor lists + TTs .. input.py:235-267, 705-896
  initial state / `any` .................... input.py:270-337
  fixed nodes (heap-ordered variations) .... input.py:340-420
  perturbations + time-step ranges ......... input.py:423-577, 664-702
  target state ............................. input.py:580-661
  problem count N = 2^a * 3^b .............. input.py:899-928
Every rejected input raises ValueError (wrapped into InputValidationException by
`process_input`), as the reference's tests/input_tests.py expects.

The implementation is independent (table-driven section parsers, a compiled-once rule
evaluator); only the observable behaviour follows the reference, including its quirks:
node-name substitution is a plain substring replacement, longest name first
(input.py:806-817), and variation lists keep `heapq` array order (input.py:401-412,551-568).
"""
import re
import heapq
import logging
import itertools
import collections.abc
from shutil import copy2

import yaml
from yaml.constructor import ConstructorError
from yaml.nodes import MappingNode

from .constants import Mode, NodeStateRange, mode_descriptions
from .model import encode_state, majority

_SECTION = 'section'
_NODE = 'node'
_PSTATE = 'perturbed node state'

_CONST_STATES = "either '0' or '1'"
_LIMITED_STATES = "either '0', '1', or 'any'"
_FULL_STATES = "either '0', '1', 'any', '0?', '1?', or 'any?'"

_RE_NONCONSTANT = re.compile(r'^(0\?|1\?|any\??)$', re.I)
_RE_ANY = re.compile(r'^any$', re.I)
_RE_MAYBE_ANY = re.compile(r'^any\?$', re.I)
_RE_RULE_DELIMITERS = re.compile(r'\(|\)|,|\b(?:and|or|not|majority)\b', re.I)
_RE_RESERVED_NAME = re.compile(r'^(:?0|1|and|or|not|majority)$', re.I)
_RE_BAD_NAME_CHARS = re.compile(r'\s|\(|\)|,')
_RE_TIME_ITEM = re.compile(r'^(\d+)(?:\s*-\s*(\d+))?$')


class InputValidationException(Exception):
    """Input file failed validation (reference: input.py:34-38)."""


class DuplicateKeyError(ConstructorError):
    """A YAML mapping holds the same key twice (reference: input.py:41-45)."""


class UniqueKeyLoader(yaml.BaseLoader):
    """BaseLoader (every scalar stays a string) that refuses duplicate mapping keys."""

    def construct_mapping(self, node, deep=False):
        if not isinstance(node, MappingNode):
            raise ConstructorError(None, None, 'expected a mapping node, but found %s' % node.id,
                                   node.start_mark)
        result = {}
        for key_node, value_node in node.value:
            key = self.construct_object(key_node, deep=deep)
            if not isinstance(key, collections.abc.Hashable):
                raise ConstructorError('while constructing a mapping', node.start_mark,
                                       'found unhashable key', key_node.start_mark)
            if key in result:
                raise DuplicateKeyError(None, None, "Duplicate key '{}' found{}".format(
                    key_node.value, key_node.start_mark))
            result[key] = self.construct_object(value_node, deep=deep)
        return result


def compile_err_msg(problem_text, location_dict):
    """'section "x", node "y":\\n<problem>' (reference: input.py:955-969)."""
    where = ', '.join('{} "{}"'.format(k, location_dict[k])
                      for k in (_SECTION, _NODE, _PSTATE) if k in location_dict)
    return where + ':\n' + problem_text if where else problem_text


def _fail(problem_text, **location):
    loc = {}
    if 'section' in location:
        loc[_SECTION] = location['section']
    if 'node' in location:
        loc[_NODE] = location['node']
    if 'pstate' in location:
        loc[_PSTATE] = location['pstate']
    raise ValueError(compile_err_msg(problem_text, loc))


def validate_raw_dict(raw_dict, key_text, location_dict):
    """Strip keys; duplicate keys (up to whitespace) are an error (reference: input.py:931-952)."""
    clean = {}
    for raw_key, value in raw_dict.items():
        key = raw_key.strip()
        if key in clean:
            raise ValueError(compile_err_msg("Duplicate {} '{}'.".format(key_text, key), location_dict))
        clean[key] = value
    return clean


# --------------------------------------------------------------------------- nodes

def parse_raw_input_node_names(raw_input_node_names):
    section = 'nodes'
    bad_format = "Expected sequence of (YAML-compliant) nodes. Example:\n'- node1\n- node2\n- node3'"
    if raw_input_node_names is None:
        raise ValueError('Nodes are missing.')
    raw = raw_input_node_names or []
    if not isinstance(raw, list):
        _fail(bad_format, section=section)
    if not raw:
        _fail('No nodes specified.', section=section)
    names = []
    for item in raw:
        if not isinstance(item, str):
            _fail(bad_format, section=section)
        names.append(item.strip())
    seen = collections.Counter(names)
    for name in names:
        if seen[name] > 1:
            _fail("Duplicate node '{}'.".format(name), section=section)
        if _RE_RESERVED_NAME.match(name):
            _fail("Node cannot be '0', '1', 'and', 'or', 'not', 'majority' (case-insensitive).",
                  section=section, node=name)
        if _RE_BAD_NAME_CHARS.search(name):
            _fail('Name cannot contain whitespaces, parentheses, or commas.', section=section, node=name)
    logging.getLogger().info('Read Boolean network of {} nodes.'.format(len(names)))
    return names


# --------------------------------------------------------------------------- update rules

_RULE_BAD_FORMAT = \
    "Bad format of update rule. Expected valid logical expression, consisting of nodes, " \
    "constants '0' and '1', operators 'and', 'or', 'not', and function 'majority(...)' " \
    "for any number of arguments. Examples: 'node5 and (not node2 or node3)', " \
    "'majority(node1, node2, not node3, node4, 1)'."


def parse_raw_input_update_rules(raw_input_update_rules, node_names):
    section = 'update rules'
    bad_format = "Expected mappings of nodes to their update rules. " \
                 "Example:\n'node1: node5 and (not node2 or node3)\n" \
                 "node2: majority(node1, node2, not node3, node4, 1)\nnode3: node3'"
    if raw_input_update_rules is None:
        raise ValueError('Update rules are missing.')
    raw = raw_input_update_rules or {}
    if not isinstance(raw, dict):
        _fail(bad_format, section=section)
    rules = validate_raw_dict(raw, _NODE, {_SECTION: section})
    return parse_input_update_rules(rules, node_names, {_SECTION: section})


def parse_predecessor_node_names_from_update_rule(update_rule):
    """Operand names of a rule = what is left between operators/parentheses/commas (input.py:832-842)."""
    return {piece.strip() for piece in _RE_RULE_DELIMITERS.split(update_rule)} - {''}


def generate_safe_node_names(node_names):
    """'node<i>' identifiers, suffixed with '_' until disjoint from the real names (input.py:845-864)."""
    taken = set(node_names)
    base = 'node'
    while True:
        safe = ['{}{}'.format(base, i) for i in range(len(node_names))]
        if taken.isdisjoint(safe):
            return safe
        base += '_'


def _check_rule_commas(update_rule, section_location):
    """Commas are legal only directly inside a majority(...) call (input.py:768-793)."""
    arg_starts = {m.end() for m in re.finditer('majority', update_rule, re.I)}
    depth = 0
    in_majority = False
    for pos, ch in enumerate(update_rule):
        if ch == '(':
            depth += 1
            if pos in arg_starts:
                in_majority = True
        elif ch == ')' and depth > 0:
            depth -= 1
            if depth == 0:
                in_majority = False
        elif ch == ',' and not (in_majority and depth == 1):
            raise ValueError(compile_err_msg('Unexpected comma.', section_location))
    if re.search(r',\s*\)', update_rule):
        raise ValueError(compile_err_msg('Last argument missing in majority function call.',
                                         section_location))


def build_truth_table_from_safe_update_rule(safe_update_rule, predecessor_nodes, safe_node_names,
                                            err_msg):
    """
    Truth table {tuple of predecessor states (in predecessor order) -> bool} of a rule already
    rewritten to Python syntax.  Same rows and same evaluation semantics (Python and/or/not,
    `majority`) as the reference (input.py:867-896); the expression is compiled once.
    """
    try:
        code = compile(safe_update_rule, '<update rule>', 'eval')
    except BaseException:
        raise ValueError(err_msg)
    names = [safe_node_names[p] for p in predecessor_nodes]
    scope = {'majority': majority}
    table = {}
    for row in itertools.product((False, True), repeat=len(names)):
        try:
            value = eval(code, scope, dict(zip(names, row)))
        except BaseException:
            raise ValueError(err_msg)
        if value not in {False, True}:
            raise ValueError(err_msg)
        table[row] = value
    return table


def parse_input_update_rules(input_update_rules, node_names, section_location_dict):
    for stray in input_update_rules.keys() - set(node_names):
        raise ValueError(compile_err_msg("Unknown node '{}'.".format(stray), section_location_dict))

    index_of = {}
    for i, name in enumerate(node_names):
        index_of.setdefault(name, i)

    predecessor_lists, rule_texts = [], []
    for name in node_names:
        if name not in input_update_rules:
            raise ValueError(compile_err_msg("Missing '{}' update rule.".format(name),
                                             section_location_dict))
        rule = input_update_rules[name]
        here = dict(section_location_dict)
        here[_NODE] = name
        if not isinstance(rule, str):
            raise ValueError(compile_err_msg(_RULE_BAD_FORMAT, here))
        preds = set()
        for operand in parse_predecessor_node_names_from_update_rule(rule):
            if operand in index_of:
                preds.add(index_of[operand])
            elif operand not in ('0', '1'):
                raise ValueError(compile_err_msg("Unknown expression '{}'.".format(operand), here))
        if re.search(r'majority(?!\()', rule, re.I):
            raise ValueError(compile_err_msg('Majority function not followed by parentheses.',
                                             section_location_dict))
        _check_rule_commas(rule, section_location_dict)
        predecessor_lists.append(sorted(preds))
        rule_texts.append(rule)

    safe_names = generate_safe_node_names(node_names)
    truth_tables = []
    for name, preds, rule in zip(node_names, predecessor_lists, rule_texts):
        text = rule
        # Plain substring replacement, longest operand first (ties keep node order).
        for p in sorted(preds, key=lambda i: len(node_names[i]), reverse=True):
            text = re.sub(re.escape(node_names[p]), safe_names[p], text)
        text = text.lower()
        text = re.sub(r'\b0\b', 'False', text)
        text = re.sub(r'\b1\b', 'True', text)
        here = dict(section_location_dict)
        here[_NODE] = name
        truth_tables.append(build_truth_table_from_safe_update_rule(
            text, preds, safe_names, compile_err_msg(_RULE_BAD_FORMAT, here)))
    return predecessor_lists, truth_tables


# --------------------------------------------------------------------------- initial state

def _dict_section(raw, section, bad_format, missing_text=None):
    """Common front part of the mapping sections: None/empty/type handling + key cleanup."""
    if raw is None and missing_text is not None:
        raise ValueError(missing_text)
    raw = raw or {}
    if not isinstance(raw, dict):
        _fail(bad_format, section=section)
    return validate_raw_dict(raw, _NODE, {_SECTION: section})


def parse_raw_input_initial_state(raw_input_initial_state, node_names):
    section = 'initial state'
    given = _dict_section(
        raw_input_initial_state, section,
        "Expected mappings of nodes to their initial states. Example:\n'node1: 0\nnode2: any\nnode3: 1'",
        'Initial state missing.')
    for stray in given.keys() - set(node_names):
        _fail("Unknown node '{}'.".format(stray), section=section)
    state = [False] * len(node_names)
    varied = []
    for node, name in enumerate(node_names):
        if name not in given:
            _fail("Missing initial '{}' state.".format(name), section=section)
        raw_state = given[name]
        if not isinstance(raw_state, str):
            _fail('Initial node state must be {}.'.format(_LIMITED_STATES), section=section, node=name)
        text = raw_state.strip()
        if text == '1':
            state[node] = True
        elif text == '0':
            pass
        elif _RE_ANY.match(text):
            varied.append(node)
        else:
            _fail("Bad initial node state '{}', must be {}.".format(text, _LIMITED_STATES),
                  section=section, node=name)
    return state, varied


# --------------------------------------------------------------------------- fixed nodes / perturbations

def _classify_state(text):
    """'0','1' -> ('const', bool); '0?','1?','any','any?' -> ('vary', NodeStateRange); else None."""
    if text == '0':
        return 'const', False
    if text == '0?':
        return 'vary', NodeStateRange.MAYBE_FALSE
    if text == '1':
        return 'const', True
    if text == '1?':
        return 'vary', NodeStateRange.MAYBE_TRUE
    if _RE_ANY.match(text):
        return 'vary', NodeStateRange.TRUE_OR_FALSE
    if _RE_MAYBE_ANY.match(text):
        return 'vary', NodeStateRange.MAYBE_TRUE_OR_FALSE
    return None


def parse_raw_input_fixed_nodes(raw_input_fixed_nodes, node_names, mode):
    section = 'fixed nodes'
    proper = _CONST_STATES if mode == Mode.ATTRACT else _FULL_STATES
    proper_ext = proper + 'in {} mode'.format(mode_descriptions[mode])
    example = "'node2: 0\nnode7: {}\nnode8: 1'".format('1' if mode == Mode.ATTRACT else 'any')
    given = _dict_section(
        raw_input_fixed_nodes, section,
        'Expected mappings of nodes to their fixed states. Example:\n{}'.format(example))
    fixed, variations = {}, []
    for name, raw_state in given.items():
        if name not in node_names:
            _fail("Unknown node '{}'.".format(name), section=section)
        node = node_names.index(name)
        if not isinstance(raw_state, str):
            _fail('Fixed node state must be {}.'.format(proper_ext), section=section, node=name)
        text = raw_state.strip()
        if mode == Mode.ATTRACT and _RE_NONCONSTANT.match(text):
            _fail("Fixed node state '{}' is forbidden in '{}' mode. Must be {}.".format(
                text.lower(), mode_descriptions[mode], proper), section=section, node=name)
        kind = _classify_state(text)
        if kind is None:
            _fail("Bad fixed node state '{}'. Must be {}.".format(text, proper_ext),
                  section=section, node=name)
        if kind[0] == 'const':
            fixed[node] = kind[1]
        else:
            if kind[1] == NodeStateRange.TRUE_OR_FALSE:
                fixed[node] = False   # origin value; digit 1 of the variation flips it
            heapq.heappush(variations, (node, kind[1]))
    return fixed, variations


def parse_raw_input_time_steps(raw_input_time_steps):
    """'1-3, 5,7' -> 1,2,3,5,7 (generator; reference: input.py:664-702)."""
    for piece in raw_input_time_steps.split(','):
        item = piece.strip()
        if not item:
            continue
        m = _RE_TIME_ITEM.match(item)
        if m is None:
            raise ValueError("'{}' is not a valid time step or interval.".format(item))
        lo = int(m.group(1))
        if m.group(2) is None:
            yield lo
            continue
        hi = int(m.group(2))
        if lo > hi:
            raise ValueError("'{}-{}' is not a valid time interval.".format(lo, hi))
        yield from range(lo, hi + 1)


def parse_raw_input_perturbations(raw_input_perturbations, node_names, mode, max_t):
    section = 'perturbations'
    attract = mode == Mode.ATTRACT
    times_bad = 'Bad format of perturbation times. Expected sequence of times or time intervals, ' \
                'separated with semicolons. Examples: 1-3,5,7, 4,6, 10.'
    sample_1 = "'{{0: 1-3,5,7, {}: 4,6}}'".format('1' if attract else 'any?')
    sample_2 = "'{{{}: 10}}'".format('1' if attract else 'any')
    node_bad = 'Bad format of node perturbations. Expected mappings of perturbed node states to' \
               ' the times at which they must occur. Examples: {}, {}.'.format(sample_1, sample_2)
    section_bad = "Expected mappings of nodes to their perturbations. Example:\n'node2: {{{}}}\nnode7: {{{}}}'" \
        .format(sample_1, sample_2)
    proper = _CONST_STATES if attract else _FULL_STATES
    proper_ext = proper + 'in {} mode'.format(mode_descriptions[mode])

    given = _dict_section(raw_input_perturbations, section, section_bad)
    by_t, variations = {}, []
    for name, raw_node_perturbations in given.items():
        if name not in node_names:
            _fail("Unknown node '{}'.".format(name), section=section)
        node = node_names.index(name)
        if not isinstance(raw_node_perturbations, dict):
            _fail(node_bad, section=section, node=name)
        per_state = validate_raw_dict(raw_node_perturbations, _PSTATE, {_SECTION: section, _NODE: name})
        for state_text, raw_times in per_state.items():
            if not isinstance(raw_times, str):
                _fail(times_bad, section=section, node=name, pstate=state_text)
            for t in parse_raw_input_time_steps(raw_times):
                if t <= 0:
                    _fail('Bad perturbation time {}. Must be positive.', section=section, node=name,
                          pstate=state_text)
                if t > max_t:
                    _fail('Bad perturbation time {}. Must not exceed simulation length {}.'.format(t, max_t),
                          section=section, node=name, pstate=state_text)
                if node in by_t.get(t, ()) or any((t, node, r) in variations for r in NodeStateRange):
                    _fail('Perturbations at time {} overlap.'.format(t), section=section, node=name)
                if attract and _RE_NONCONSTANT.match(state_text):
                    _fail("Perturbed node state '{}' is forbidden in '{}' mode. Must be {}.".format(
                        state_text.lower(), mode_descriptions[mode], proper), section=section, node=name)
                kind = _classify_state(state_text)
                if kind is None:
                    _fail("Bad perturbed node state '{}'. Must be {}.".format(state_text, proper_ext),
                          section=section, node=name)
                if kind[0] == 'const':
                    by_t.setdefault(t, {})[node] = kind[1]
                else:
                    if kind[1] == NodeStateRange.TRUE_OR_FALSE:
                        by_t.setdefault(t, {})[node] = False
                    heapq.heappush(variations, (t, node, kind[1]))
    return by_t, variations


# --------------------------------------------------------------------------- target state

def parse_raw_input_target_state(raw_input_target_state, node_names, mode):
    section = 'target state'
    if mode != Mode.TARGET:
        if raw_input_target_state is not None:
            logging.getLogger().warning(
                "Target state is only used in '{}' mode.".format(mode_descriptions[Mode.TARGET]))
        return None, None
    given = _dict_section(
        raw_input_target_state, section,
        "Expected mappings of nodes to their target states. Example:\n'node1: 0\nnode2: 1\nnode3: any'",
        'Target state missing.')
    for stray in given.keys() - set(node_names):
        _fail("Unknown node '{}'.".format(stray), section=section)
    wanted = [False] * len(node_names)
    constrained = set(range(len(node_names)))
    for node, name in enumerate(node_names):
        if name not in given:
            _fail("Missing target '{}' state.".format(name), section=section)
        raw_state = given[name]
        if not isinstance(raw_state, str):
            _fail('Target node state must be {}.'.format(_LIMITED_STATES), section=section, node=name)
        text = raw_state.strip()
        if text == '0':
            pass
        elif text == '1':
            wanted[node] = True
        elif _RE_ANY.match(text):
            constrained.discard(node)
        else:
            _fail("Bad target node state '{}'. Must be {}.".format(text, _LIMITED_STATES),
                  section=section, node=name)
    _, substate_code = encode_state(constrained, wanted)
    return substate_code, constrained


# --------------------------------------------------------------------------- whole file

def count_simulation_problems(initial_state_variations, fixed_node_variations, perturbation_variations):
    """N = 2^(#binary digits) * 3^(#ternary digits)  (reference: input.py:899-928)."""
    ternary = sum(1 for _, r in fixed_node_variations if r == NodeStateRange.MAYBE_TRUE_OR_FALSE) + \
        sum(1 for _, _, r in perturbation_variations if r == NodeStateRange.MAYBE_TRUE_OR_FALSE)
    binary = len(initial_state_variations) + len(fixed_node_variations) + \
        len(perturbation_variations) - ternary
    return 2 ** binary * 3 ** ternary


_VALID_SECTIONS = ('nodes', 'update rules', 'initial state', 'fixed nodes', 'perturbations',
                   'target state')


def parse_input_data(input_data, max_t, mode):
    """Validate an already-loaded YAML mapping; returns the reference's configuration dict."""
    cfg = {}
    cfg['node names'] = parse_raw_input_node_names(input_data.get('nodes'))
    cfg['incoming node lists'], cfg['truth tables'] = parse_raw_input_update_rules(
        input_data.get('update rules'), cfg['node names'])
    initial_state, initial_state_variations = parse_raw_input_initial_state(
        input_data.get('initial state'), cfg['node names'])
    fixed_nodes, fixed_nodes_variations = parse_raw_input_fixed_nodes(
        input_data.get('fixed nodes'), cfg['node names'], mode)
    perturbed_nodes_by_t, perturbation_variations = parse_raw_input_perturbations(
        input_data.get('perturbations'), cfg['node names'], mode, max_t)
    cfg['target substate code'], cfg['target node set'] = parse_raw_input_target_state(
        input_data.get('target state'), cfg['node names'], mode)
    for section_name in input_data:
        if section_name not in _VALID_SECTIONS:
            logging.getLogger().warning("Unknown section '{}'.".format(section_name))
    cfg['origin simulation problem'] = (initial_state, fixed_nodes, perturbed_nodes_by_t)
    cfg['simulation problem variations'] = (initial_state_variations, fixed_nodes_variations,
                                            perturbation_variations)
    cfg['total combination count'] = count_simulation_problems(
        initial_state_variations, fixed_nodes_variations, perturbation_variations)
    return cfg


def parse_input(file_name, max_t, mode):
    with open(file_name) as stream:
        input_data = yaml.load(stream, Loader=UniqueKeyLoader)
    return parse_input_data(input_data, max_t, mode)


def parse_input_text(text, max_t, mode):
    """Same as parse_input for YAML text already in memory (used by bench.py and the tests)."""
    return parse_input_data(yaml.load(text, Loader=UniqueKeyLoader), max_t, mode)


def process_input(input_file, output_directory, max_t, mode, copy_input=True):
    """Copy the input next to the outputs, parse, map validation failures (reference: input.py:75-95).
    copy_input: in a multi-rank job only the rank that owns the output directory copies."""
    if copy_input:
        try:
            copy2(input_file, output_directory)
        except IOError as e:
            logging.getLogger().warning('Failed to copy input file into the output directory: {}'.format(e))
    try:
        return parse_input(input_file, max_t, mode)
    except (ValueError, KeyError, DuplicateKeyError) as e:
        logging.getLogger().error('Input validation failed: {}'.format(e))
        raise InputValidationException('Failed to validate input.')
