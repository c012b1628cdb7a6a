"""Writes swizzles.inc: the union members that give glsl::vec its .xy / .rgb / .wzyx ... swizzles (xyzw and rgba sets)."""
import itertools, os
out = []
for n in (2, 3, 4):
    members = []
    for names in ("xyzw"[:n], "rgba"[:n]):
        for k in (2, 3, 4):
            for idx in itertools.product(range(n), repeat=k):
                members.append("Swz<T, vec<T, %d>, %d, %s> %s;" % (k, n, ", ".join(map(str, idx)), "".join(names[i] for i in idx)))
    out.append("#define GLSL_SWZ%d(T) %s" % (n, " ".join(members)))
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "swizzles.inc"), "w").write("\n".join(out) + "\n")
