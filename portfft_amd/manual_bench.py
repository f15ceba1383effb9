"""Descriptor strings of the reference's manual benchmark ("d=cpx,n=1024x1024,b=8,s=split,p=ip").

Same grammar, keys, value spellings and error texts as `test/bench/portfft/register_manual_bench.hpp`:
keys and their short forms (`:36-40`), `key=value` tokens separated by ',' with every key at most once and an empty
token ending the string (`get_arg_map`, `:77-113`), positive integers and 'x'-separated lists (`:128-161`), the fields
a token sets (`fill_descriptor`, `:163-211`), `domain` and `lengths` mandatory (`:218-236`).  `bench.py --manual`
feeds the result to the engine; host-only logic, covered by tests/test_manual_bench.py.
"""
from . import api

ARG_KEYS = (
    ("domain", "d"), ("lengths", "n"), ("batch", "b"), ("fwd_strides", "fs"), ("bwd_strides", "bs"),
    ("fwd_dist", "fd"), ("bwd_dist", "bd"), ("scale", "sx"), ("storage", "s"), ("placement", "p"),
)
_LONG = {long: long for long, _ in ARG_KEYS}
_LONG.update({short: long for long, short in ARG_KEYS})


class bench_error(RuntimeError):
    pass


class invalid_value(bench_error):
    def __init__(self, key, value):
        super().__init__("Invalid '%s' value: '%s'" % (key, value))


def get_arg_map(arg):
    """{key as written: value}; long and short spellings of one key are distinct entries, like the reference's map
    (the long one wins in get_arg)."""
    arg_map = {}
    for token in arg.split(","):
        if token == "":
            break  # the reference stops at the first empty token
        if "=" not in token:
            raise bench_error("Invalid token '%s'" % token)
        key, value = token.split("=", 1)
        if key in arg_map:
            raise bench_error("Key can only be specified once: '%s'" % key)
        if key not in _LONG:
            raise bench_error("Invalid key: '%s'" % key)
        if value == "":
            raise invalid_value(key, value)
        arg_map[key] = value
    return arg_map


def get_arg(arg_map, long_key):
    short = dict(ARG_KEYS)[long_key]
    return arg_map.get(long_key, arg_map.get(short, ""))


def get_unsigned(key, value):
    size = _stol_prefix(value)
    if size is None or size <= 0:
        raise bench_error("Invalid '%s' value: '%s' must be a positive integer" % (key, value))
    return size


def _stol_prefix(value):
    """std::stol: leading whitespace, optional sign, digits; a numeric prefix is enough ("12abc" -> 12)"""
    s = value.lstrip()
    i = 1 if s[:1] in "+-" else 0
    j = i
    while j < len(s) and s[j].isdigit():
        j += 1
    return int(s[:j]) if j > i else None


def get_vec_unsigned(key, value):
    vec = []
    for token in value.split("x"):
        if token == "":
            break
        vec.append(get_unsigned(key, token))
    return vec


_STORAGE = {"complex": 0, "cpx": 0, "interleaved": 0, "int": 0, "real_real": 1, "rr": 1, "split": 1, "sp": 1}
_PLACEMENT = {"in_place": 0, "ip": 0, "out_of_place": 1, "oop": 1}
_DOMAIN = {"complex": api.domain.COMPLEX, "cpx": api.domain.COMPLEX, "real": api.domain.REAL, "re": api.domain.REAL}


def fill_descriptor(arg_map, desc):
    arg = get_arg(arg_map, "batch")
    if arg:
        desc.number_of_transforms = get_unsigned("batch", arg)
    arg = get_arg(arg_map, "fwd_strides")
    if arg:
        desc.forward_strides = get_vec_unsigned("fwd_strides", arg)
    arg = get_arg(arg_map, "bwd_strides")
    if arg:
        desc.backward_strides = get_vec_unsigned("bwd_strides", arg)
    arg = get_arg(arg_map, "fwd_dist")
    if arg:
        desc.forward_distance = get_unsigned("fwd_dist", arg)
    arg = get_arg(arg_map, "bwd_dist")
    if arg:
        desc.backward_distance = get_unsigned("bwd_dist", arg)
    arg = get_arg(arg_map, "scale")
    if arg:
        try:
            scale = float(arg)
        except ValueError:
            raise invalid_value("scale", arg)
        desc.forward_scale = scale
        desc.backward_scale = scale
    arg = get_arg(arg_map, "storage")
    if arg:
        if arg not in _STORAGE:
            raise invalid_value("storage", arg)
        desc.complex_storage = api.complex_storage(_STORAGE[arg])
    arg = get_arg(arg_map, "placement")
    if arg:
        if arg not in _PLACEMENT:
            raise invalid_value("placement", arg)
        desc.placement = api.placement(_PLACEMENT[arg])


def descriptor_from_string(desc_str, scalar="f32"):
    """the descriptor `bench_manual_float` / `bench_manual_double` would build from `desc_str`"""
    arg_map = get_arg_map(desc_str)
    domain_str = get_arg(arg_map, "domain")
    if domain_str == "":
        raise bench_error("'domain' must be specified")
    if domain_str not in _DOMAIN:
        raise invalid_value("domain", domain_str)
    lengths = get_vec_unsigned("lengths", get_arg(arg_map, "lengths"))
    if not lengths:
        raise bench_error("'lengths' must be specified")
    desc = api.descriptor(lengths, scalar, _DOMAIN[domain_str])
    fill_descriptor(arg_map, desc)
    return desc


def help_text(prog):
    w = 25
    rows = [
        ("domain", "complex|cpx|real|re (mandatory)"), ("lengths", "AxBx... positive integers (mandatory)"),
        ("batch", "number of transforms"), ("fwd_strides", "AxBx... forward strides"),
        ("bwd_strides", "AxBx... backward strides"), ("fwd_dist", "forward distance"), ("bwd_dist", "backward distance"),
        ("scale", "forward and backward scale"), ("storage", "complex|cpx|interleaved|int or real_real|rr|split|sp"),
        ("placement", "in_place|ip or out_of_place|oop"),
    ]
    short = dict(ARG_KEYS)
    lines = ["usage: %s --manual key=value[,key=value...] [--precision float|double]" % prog, "keys:"]
    for key, what in rows:
        lines.append(("\t'%s', '%s'" % (key, short[key])).ljust(w) + what)
    return "\n".join(lines)
