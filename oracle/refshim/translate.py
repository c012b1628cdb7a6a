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


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    return re.sub(r"//[^\n]*", "", t)


def expand(path, depth=0):
    text = strip_comments(open(path).read())
    out = []
    for line in text.split("\n"):
        m = re.match(r'\s*#\s*include\s+"([^"]+)"', line)
        if m:
            out.append(expand(os.path.normpath(os.path.join(os.path.dirname(path), m.group(1))), depth + 1))
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


def translate(rel, defines=()):
    t = expand(os.path.join(REF_SHADERS, rel))
    for pat, rep in TOKEN_PATCHES.get(rel, []) + TOKEN_PATCHES["*"]:
        t = re.sub(pat, rep, t)
    regs = []

    # workgroup size
    def local_size(m):
        d = dict((k.strip(), v.strip()) for k, v in (kv.split("=") for kv in m.group(1).split(",")))
        return "\n".join("#define SHADER_LOCAL_%s (%s)" % (a.upper(), d.get("local_size_" + a, "1")) for a in "xyz")
    t, n_local = re.subn(r"layout\s*\(([^)]*local_size_x[^)]*)\)\s*in\s*;", local_size, t)
    if not n_local:
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
                regs.append(mn)
            return "\n".join(out)
        members = []
        for mm in re.finditer(r"(\w+)\s+(\w+)\s*((?:\[[^\]]*\])*)\s*;", body):
            ty, mn, dims = mm.groups()
            if dims.replace(" ", "") == "[]":
                members.append("%s* %s;" % (ty, mn))
            else:
                members.append("%s %s%s;" % (ty, mn, dims))
            if not inst_dims:
                regs.append("%s.%s" % (inst, mn))
        regs.append(inst)
        if inst_dims and inst_dims.replace(" ", "") == "[]":
            inst_dims = "[1024]"   # unsized descriptor arrays: a fixed table
        return "struct %s_block { %s };\nstatic %s_block %s%s;" % (name, " ".join(members), name, inst, inst_dims or "")
    t = re.sub(r"layout\s*\(([^)]*)\)\s*(?:(?:readonly|writeonly|restrict|coherent)\s+)*(?:uniform|buffer)\s+(\w+)\s*\{([^}]*)\}\s*(\w+)?\s*(\[[^\]]*\])?\s*;", block, t)

    # opaque resources
    def resource(m):
        ty, name, arr = m.group(1), m.group(2), m.group(3)
        regs.append(name)
        return ("static %s* %s;" if arr else "static %s %s;") % (ty, name)
    t = re.sub(r"layout\s*\([^)]*\)\s*uniform\s+(?:(?:readonly|writeonly|restrict|coherent)\s+)*(%s)\s+(\w+)\s*(\[\s*\])?\s*;" % "|".join(RESOURCE_TYPES), resource, t)

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

    name = re.sub(r"\W", "_", os.path.splitext(os.path.basename(rel))[0]) + "".join("_" + re.sub(r"\W", "_", d.split("=")[0]) for d in defines)
    head = ['// GENERATED by oracle/refshim/translate.py from the reference shader %s — do not commit.' % rel, '#include "glsl.h"', "#undef M_PI"]
    for d in defines:
        k, _, v = d.partition("=")
        head.append("#define %s %s" % (k, v))
    reg_lines = ",\n".join('    { "%s", (void*)&%s, sizeof(%s) }' % (r, r, r) for r in regs)
    tail = "\nstatic const Reg g_regs[] = {\n%s\n};\n#include \"runtime.inc\"\n} // namespace glsl\n" % reg_lines
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, name + ".cpp")
    with open(out, "w") as f:
        f.write("\n".join(head) + "\nnamespace glsl {\n" + t + tail)
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    defs = []
    while "-D" in args:
        i = args.index("-D")
        defs.append(args[i + 1])
        del args[i:i + 2]
    print(translate(args[0], defs))
