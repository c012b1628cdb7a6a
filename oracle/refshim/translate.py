"""ORACLE — TEST INFRASTRUCTURE ONLY.

GLSL -> C++ text translation for oracle/refshim/glsl.h.  Reads ONE shader of the reference where it lies
(/root/reference/src/shaders/<rel>), expands its #include directives, rewrites the handful of constructs C++ cannot
parse (layout declarations, interface blocks, parameter qualifiers, array constructors, unsuffixed float literals) and
writes the result to oracle/_ref/gen/<name>.cpp.  The translated text is reference code: it is never written anywhere
but oracle/_ref/ (git-ignored), and this script contains none of it — only syntax rules.

    python oracle/refshim/translate.py shadows/shadows_denoise_atrous.comp [-D NAME[=VALUE] ...]
"""
import os
import re
import sys

REF_SHADERS = "/root/reference/src/shaders"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "_ref", "gen")

RESOURCE_TYPES = ("sampler2D", "usampler2D", "samplerCube", "image2D", "uimage2D", "accelerationStructureEXT")
# spots where GLSL is laxer than C++ (scalar swizzles, names that are C++ keywords ...): pure token substitutions
TOKEN_PATCHES = {
    "shadows/shadows_denoise_atrous.comp": [(r"\bvar\.r\b", "var")],          # .r of a scalar
    "reflections/reflections_denoise_atrous.comp": [(r"\bvar\.r\b", "var")],
    "*": [(r"\bbase_grid_coord\s*\(", "base_grid_coord_fn("), (r"\bsrand\s*\(", "srand_fn(")],               # a local variable shadows the function it is initialised from
}


# source-level variants, applied BEFORE comments are stripped: { variant: [(file suffix, regex, replacement)] }
VARIANTS = {
    # SURVEY.md section 8f row 3's optional extension: the recursive traceRayEXT the reference ships commented out in
    # ground_truth_path_trace.rchit:95-105, un-commented (the lines are the reference's own)
    "bounces": [("ground_truth_path_trace.rchit", r"(?m)^(\s*)//(traceRayEXT\(.*|\s{12}\S.*)$", r"\1\2")],
}


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    return re.sub(r"//[^\n]*", "", t)


def expand(path, seen=None, variant=None):
    """the file with its #include directives expanded; include guards become include-once bookkeeping here, so that
    several stages of a pipeline can be translated into one C++ file"""
    seen = set() if seen is None else seen
    if path in seen:
        return ""
    seen.add(path)
    text = open(path).read()
    for suffix, pat, rep in VARIANTS.get(variant, []):
        if path.endswith(suffix):
            text = re.sub(pat, rep, text)
    text = strip_comments(text)
    g = re.match(r"\s*#\s*ifndef\s+(\w+)\s*\n\s*#\s*define\s+\1[ \t]*\n", text)
    if g and text.rstrip().endswith("#endif"):
        text = text[g.end():text.rstrip().rfind("#endif")]
    out = []
    for line in text.split("\n"):
        m = re.match(r'\s*#\s*include\s+"([^"]+)"', line)
        if m:
            out.append(expand(os.path.normpath(os.path.join(os.path.dirname(path), m.group(1))), seen, variant))
        elif re.match(r"\s*#\s*(version|extension)\b", line):
            continue
        else:
            out.append(line)
    return "\n".join(out)


def match_paren(t, i):
    """index of the ')' matching the '(' at t[i]"""
    depth = 0
    for j in range(i, len(t)):
        if t[j] == "(":
            depth += 1
        elif t[j] == ")":
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced parentheses")


def translate_body(rel, regs, stage=None, variant=None):
    """(translated text, has_local_size); regs collects the (registry name, C++ lvalue) pairs"""
    t = expand(os.path.join(REF_SHADERS, rel), variant=variant)
    for pat, rep in TOKEN_PATCHES.get(rel, []) + TOKEN_PATCHES["*"]:
        t = re.sub(pat, rep, t)
    pre = "" if stage is None else "st%d::" % stage
    if stage is not None:
        # ray-tracing interface variables (see glsl.h: ray-tracing pipelines)
        t = re.sub(r"layout\s*\(\s*location\s*=\s*(\d+)\s*\)\s*rayPayloadEXT\s+(\w+)\s+(\w+)\s*;",
                   lambda m: "static %s %s; static int _rp%s = rt_register_payload(%d, %s, &%s, sizeof(%s));" % (m.group(2), m.group(3), m.group(1), stage, m.group(1), m.group(3), m.group(3)), t)
        t = re.sub(r"layout\s*\(\s*location\s*=\s*(\d+)\s*\)\s*rayPayloadInEXT\s+(\w+)\s+(\w+)\s*;",
                   lambda m: "static %s %s; static int _ri%s = rt_register_incoming(%d, &%s, sizeof(%s));" % (m.group(2), m.group(3), m.group(1), stage, m.group(3), m.group(3)), t)
        t = re.sub(r"hitAttributeEXT\s+vec2\s+(\w+)\s*;", r"#define \1 g_hit_attribs", t)

    # workgroup size
    def local_size(m):
        d = dict((k.strip(), v.strip()) for k, v in (kv.split("=") for kv in m.group(1).split(",")))
        return "\n".join("#define SHADER_LOCAL_%s (%s)" % (a.upper(), d.get("local_size_" + a, "1")) for a in "xyz")
    t, n_local = re.subn(r"layout\s*\(([^)]*local_size_x[^)]*)\)\s*in\s*;", local_size, t)
    if not n_local and stage is None:
        t += "\n#define SHADER_LOCAL_X 1\n#define SHADER_LOCAL_Y 1\n#define SHADER_LOCAL_Z 1\n"

    # interface blocks (uniform / buffer / push constants)
    def block(m):
        name, body, inst, inst_dims = m.group(2), m.group(3), m.group(4), m.group(5)
        if not inst:
            # no instance name: the members are globals
            out = []
            for mm in re.finditer(r"(\w+)\s+(\w+)\s*((?:\[[^\]]*\])*)\s*;", body):
                ty, mn, dims = mm.groups()
                out.append("static %s %s%s;" % (ty, mn, dims))
                regs.append((mn, pre + mn))
            return "\n".join(out)
        members = []
        for mm in re.finditer(r"(\w+)\s+(\w+)\s*((?:\[[^\]]*\])*)\s*;", body):
            ty, mn, dims = mm.groups()
            if dims.replace(" ", "") == "[]":
                members.append("%s* %s;" % (ty, mn))
            else:
                members.append("%s %s%s;" % (ty, mn, dims))
            if not inst_dims:
                regs.append(("%s.%s" % (inst, mn), "%s%s.%s" % (pre, inst, mn)))
        regs.append((inst, pre + inst))
        if inst_dims and inst_dims.replace(" ", "") == "[]":
            inst_dims = "[1024]"   # unsized descriptor arrays: a fixed table
        return "struct %s_block { %s };\nstatic %s_block %s%s;" % (name, " ".join(members), name, inst, inst_dims or "")
    t = re.sub(r"layout\s*\(([^)]*)\)\s*(?:(?:readonly|writeonly|restrict|coherent)\s+)*(?:uniform|buffer)\s+(\w+)\s*\{([^}]*)\}\s*(\w+)?\s*(\[[^\]]*\])?\s*;", block, t)

    # opaque resources
    def resource(m):
        ty, name, arr = m.group(1), m.group(2), m.group(3)
        regs.append((name, pre + name))
        return ("static %s* %s;" if arr else "static %s %s;") % (ty, name)
    t = re.sub(r"layout\s*\([^)]*\)\s*uniform\s+(?:(?:readonly|writeonly|restrict|coherent)\s+)*(%s)\s+(\w+)\s*(\[\s*\])?\s*;" % "|".join(RESOURCE_TYPES), resource, t)

    # fragment-shader interface variables: plain statics the harness writes / reads around every invocation
    def io_var(m):
        regs.append((m.group(2), pre + m.group(2)))
        return "static %s %s;" % (m.group(1), m.group(2))
    t = re.sub(r"layout\s*\(\s*location\s*=\s*\d+\s*\)\s*(?:in|out)\s+(\w+)\s+(\w+)\s*;", io_var, t)

    t = re.sub(r"\bshared\s+", "static ", t)
    # parameter qualifiers
    t = re.sub(r"\b(?:const\s+)?in\s+(\w+)\s+(\w+)\s*(?=[,)\n])", r"\1 \2", t)
    t = re.sub(r"\b(?:out|inout)\s+(\w+)\s+(\w+)\s*(?=[,)\n])", r"\1& \2", t)
    # array constructors  T[](a, b, c)  ->  { a, b, c }
    while True:
        m = re.search(r"\b\w+\s*\[\s*\d*\s*\]\s*\(", t)
        if not m:
            break
        j = match_paren(t, m.end() - 1)
        t = t[:m.start()] + "{" + t[m.end():j] + "}" + t[j + 1:]
    # float literals: GLSL's are fp32
    t = re.sub(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.])", r"\1f", t)
    t = re.sub(r"\bvoid\s+main\s*\(\s*\)", "void shader_main()", t)
    return t


def _write(name, rel_desc, defines, body, regs, runtime):
    head = ['// GENERATED by oracle/refshim/translate.py from the reference shader(s) %s — do not commit.' % rel_desc, '#include "glsl.h"', "#undef M_PI"]
    for d in defines:
        k, _, v = d.partition("=")
        head.append("#define %s %s" % (k, v))
    reg_lines = ",\n".join('    { "%s", (void*)&%s, sizeof(%s) }' % (n, lv, lv) for n, lv in regs)
    tail = "\nstatic const Reg g_regs[] = {\n%s\n};\n%s\n} // namespace glsl\n" % (reg_lines, runtime)
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, name + ".cpp")
    with open(out, "w") as f:
        f.write("\n".join(head) + "\nnamespace glsl {\n" + body + tail)
    return out


def translate(rel, defines=()):
    regs = []
    t = translate_body(rel, regs)
    name = re.sub(r"\W", "_", os.path.splitext(os.path.basename(rel))[0]) + "".join("_" + re.sub(r"\W", "_", d.split("=")[0]) for d in defines)
    return _write(name, rel, defines, t, regs, '#include "runtime.inc"')


def translate_pipeline(name, stages, defines=(), variant=None):
    """stages: [(rel, kind)] with kind 0 = ray generation (first), 1 = closest hit, 2 = miss (in SBT order per kind).
    Every stage is translated into its own namespace st<i>; registry names are '<i>:<glsl name>'."""
    regs, body, table = [], [], []
    for i, (rel, kind) in enumerate(stages):
        sregs = []
        t = translate_body(rel, sregs, stage=i, variant=variant)
        macros = sorted(set(re.findall(r"^\s*#\s*define\s+(\w+)", t, flags=re.M)))
        body.append("namespace st%d {\n%s\n}\n%s\n" % (i, t, "\n".join("#undef " + m for m in macros)))
        regs += [("%d:%s" % (i, n), lv) for n, lv in sregs]
        table.append("{ st%d::shader_main, %d }" % (i, kind))
    runtime = ("#define SHADER_LOCAL_X 1\n#define SHADER_LOCAL_Y 1\n#define SHADER_LOCAL_Z 1\nstatic void shader_main() {}\n#include \"runtime.inc\"\n"
               "#define RT_STAGE_TABLE { %s }\n#include \"runtime_rt.inc\"" % ", ".join(table))
    return _write(name, " + ".join(r for r, _ in stages), defines, "\n".join(body), regs, runtime)


if __name__ == "__main__":
    args = sys.argv[1:]
    defs = []
    while "-D" in args:
        i = args.index("-D")
        defs.append(args[i + 1])
        del args[i:i + 2]
    print(translate(args[0], defs))
