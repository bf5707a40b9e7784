#!/usr/bin/env python3
"""Line splitter for the dense C++ / HIP sources under blockmaze_amd/csrc: every physical line longer than LIMIT columns is rewritten as
one statement per line (split at the `;`, `{` and `}` that stand outside parentheses), its trailing comment moved in front of it, and what is still
too long wrapped at a `, ` or an operator.  Only white space and the position of comments change: the token stream the compiler sees is the same
(tools/fmt_check.sh compares the device assembly and the host objects before and after).

    python tools/fmt_lines.py [--limit 160] [--check] file...

Lines it does not understand (a statement that continues on the next line, preprocessor lines, block comments that span lines) are left alone."""
import sys

LIMIT = 160
BLOCK_WORDS = ("else", "do", "try", "const", "mutable", "noexcept", "override")
WRAP_RANKS = (("; ", "{ "), (", ",), (" && ", " || ", " ? ", " : ", " + ", " - ", " = ", " | ", " ^ ", " << "))


def code_and_comment(text):
    """-> (code, comment or None, ok).  Splits at the first `//` that is outside string / character literals and /* */ comments."""
    i, n, quote = 0, len(text), None
    while i < n:
        c = text[i]
        if quote:
            if c == "\\":
                i += 2
                continue
            if c == quote:
                quote = None
        elif c in "\"'":
            if c == "'" and i > 0 and text[i - 1].isdigit() and i + 1 < n and text[i + 1].isdigit():   # a digit separator (1'000), not a character literal
                i += 1
                continue
            quote = c
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            if j < 0:
                return text, None, False
            i = j + 2
            continue
        elif text.startswith("//", i):
            return text[:i].rstrip(), text[i + 2:].strip(), True
        i += 1
    return text.rstrip(), None, quote is None


def wrap_comment(comment, indent, limit):
    words, out, cur = comment.split(), [], ""
    pre = " " * indent + "// "
    for w in words:
        if cur and len(pre) + len(cur) + 1 + len(w) > limit:
            out.append(pre + cur)
            cur = w
        else:
            cur = w if not cur else cur + " " + w
    if cur:
        out.append(pre + cur)
    return out


def split_statements(code):
    """-> list of (relative depth, text) or None when the line is not self-contained (open parenthesis at its end)."""
    pieces, cur, depth, paren, quote = [], "", 0, 0, None
    stack = []          # 'b' block brace, 'd' block brace of a do-loop, 'e' expression brace
    i, n = 0, len(code)

    def flush():
        nonlocal cur
        t = cur.strip()
        if t.startswith("}"):
            t = "} " + t[1:].lstrip() if t[1:].lstrip()[:1] not in ("", ";", ",", ")", "(") else "}" + t[1:].lstrip()
        if t:
            pieces.append((depth, t))
        cur = ""

    while i < n:
        c = code[i]
        if quote:
            cur += c
            if c == "\\" and i + 1 < n:
                cur += code[i + 1]
                i += 2
                continue
            if c == quote:
                quote = None
            i += 1
            continue
        if c in "\"'":
            if c == "'" and cur and cur[-1].isdigit() and i + 1 < n and code[i + 1].isdigit():
                cur += c
                i += 1
                continue
            quote = c
            cur += c
            i += 1
            continue
        if code.startswith("/*", i):
            j = code.find("*/", i + 2)
            if j < 0:
                return None
            cur += code[i:j + 2]
            i = j + 2
            continue
        if c in "([":
            paren += 1
        elif c in ")]":
            paren -= 1
            if paren < 0:
                return None
        elif c == "{":
            before = cur.rstrip()
            last_word = before.split()[-1] if before.split() else ""
            is_block = paren == 0 and (before == "" or before[-1] in ")]" or last_word in BLOCK_WORDS or before[-1] in "{};")
            if is_block:
                stack.append("d" if last_word == "do" else "b")
                cur = before + (" {" if before else "{")
                flush()
                depth += 1
                i += 1
                continue
            stack.append("e")
            paren += 1
        elif c == "}":
            if stack and stack[-1] == "e":
                stack.pop()
                paren -= 1
            elif paren == 0:
                was_do = bool(stack) and stack[-1] == "d"
                if stack:
                    stack.pop()
                flush()
                depth -= 1
                rest = code[i + 1:].lstrip()
                glue = rest[:1] in (";", ",", ")", "(") or rest.startswith("else") and not rest[4:5].isalnum() and rest[4:5] != "_" or was_do and rest.startswith("while")
                cur = "}"
                if not glue:
                    flush()
                elif rest[:1] not in (";", ",", ")", "("):
                    cur += " "
                i += 1
                continue
            else:
                return None
        elif c == ";" and paren == 0:
            cur += c
            flush()
            i += 1
            continue
        cur += c
        i += 1
    if paren != 0 or quote:
        return None
    flush()
    return pieces


def wrap_piece(text, indent, limit):
    """Break one statement that is still too long after a `, ` or an operator (outside string literals)."""
    out, pre = [], " " * indent
    while len(pre) + len(text) > limit:
        room = limit - len(pre)
        last, quote, i = [-1, -1, -1], None, 0
        while i < min(len(text), room):
            c = text[i]
            if quote:
                if c == "\\":
                    i += 2
                    continue
                if c == quote:
                    quote = None
            elif c == '"' or c == "'" and not (i > 0 and text[i - 1].isdigit()):
                quote = c
            elif text.startswith("/*", i):
                j = text.find("*/", i + 2)
                i = j + 2 if j >= 0 else len(text)
                continue
            else:
                for rank, seps in enumerate(WRAP_RANKS):
                    for w in seps:
                        if text.startswith(w, i) and i + len(w) <= room and i > 8:
                            last[rank] = i + len(w)
            i += 1
        # a statement boundary inside a lambda body beats a comma, a comma beats an operator — unless that leaves the line less than half full
        best = last[0] if last[0] > room // 2 else last[1] if last[1] > room * 2 // 5 else max(last)
        if best <= 0:
            break
        out.append(pre + text[:best].rstrip())
        text = text[best:].lstrip()
        pre = " " * (indent + 4)
    out.append(pre + text)
    return out


def reflow_comment_run(run, indent, limit):
    """run: the texts after `//` of consecutive comment-only lines with one indentation, at least one of them too long.  Lines that belong to one paragraph (a full line
    followed by a line that starts at the left edge) are joined and broken again; a line that starts further right (a table row, a hanging indent) keeps its column."""
    pre0 = " " * indent + "//"
    lead = [len(t) - len(t.lstrip(" ")) for t in run]
    paragraphs = []                                     # [first line index, last line index]
    for k, t in enumerate(run):
        prev_len = len(pre0) + len(run[k - 1]) if k else 0
        if k == 0 or lead[k] >= 2 or lead[k - 1] >= 2 or not t.strip() or not run[k - 1].strip() or prev_len < limit - 40:
            paragraphs.append([k, k])
        else:
            paragraphs[-1][1] = k
    out = []
    for a, b in paragraphs:
        if all(len(pre0) + len(run[k]) <= limit for k in range(a, b + 1)):
            out.extend((pre0 + run[k]).rstrip() for k in range(a, b + 1))
            continue
        first = pre0 + " " * max(1, lead[a]) if run[a].strip() else pre0
        hang = lead[b + 1] if b + 1 < len(run) and lead[b + 1] > lead[a] >= 2 else lead[a]
        cont = pre0 + " " * max(1, hang)
        words, cur, pre = " ".join(run[k].strip() for k in range(a, b + 1)).split(), "", first
        for w in words:
            if cur and len(pre) + len(cur) + 1 + len(w) > limit:
                out.append(pre + cur)
                cur, pre = w, cont
            else:
                cur = w if not cur else cur + " " + w
        if cur:
            out.append(pre + cur)
    return out


def emit_before_pragma(out, comment_lines):
    """a comment that moves in front of its statement goes in front of the statement's #pragma line too"""
    k = len(out)
    while k > 0 and out[k - 1].lstrip().startswith("#pragma"):
        k -= 1
    out[k:k] = comment_lines


def reformat(lines, limit):
    out, in_block_comment, in_macro = [], False, False
    src = [raw.rstrip("\n") for raw in lines]
    n, i = len(src), 0
    while i < n:
        line = src[i]
        stripped = line.lstrip()
        continued = in_macro
        in_macro = line.endswith("\\")
        if in_block_comment:
            out.append(line)
            if "*/" in line:
                in_block_comment = False
            i += 1
            continue
        indent = len(line) - len(stripped)
        if stripped.startswith("//") and not continued and not in_macro:
            # a run of comment-only lines with this indentation
            j = i
            while j < n and src[j].lstrip().startswith("//") and len(src[j]) - len(src[j].lstrip()) == indent and not src[j].endswith("\\"):
                j += 1
            run = [src[k].lstrip()[2:] for k in range(i, j)]
            if any(len(src[k]) > limit for k in range(i, j)):
                out.extend(reflow_comment_run(run, indent, limit))
            else:
                out.extend(src[i:j])
            i = j
            continue
        i += 1
        if continued or in_macro or stripped.startswith("#") or len(line) <= limit:
            out.append(line)
            code, _, ok = code_and_comment(stripped)
            if not ok and "/*" in stripped:
                in_block_comment = True
            continue
        code, comment, ok = code_and_comment(stripped)
        if not ok:
            out.append(line)
            if "/*" in stripped:
                in_block_comment = True
            continue
        pieces = split_statements(code)
        if pieces is None:                                 # part of a statement that spans lines: no statement split, but a break after a `, ` is always safe
            if comment:
                emit_before_pragma(out, wrap_comment(comment, indent, limit))
            out.extend(wrap_piece(code, indent, limit))
            continue
        if comment:
            emit_before_pragma(out, wrap_comment(comment, indent, limit))
        d0 = pieces[0][0]
        for d, text in pieces:
            ind = max(0, indent + 2 * (d - d0))
            out.extend(wrap_piece(text, ind, limit))
    return out


def main():
    args, limit, check = sys.argv[1:], LIMIT, False
    files = []
    while args:
        a = args.pop(0)
        if a == "--limit":
            limit = int(args.pop(0))
        elif a == "--check":
            check = True
        else:
            files.append(a)
    changed = 0
    for path in files:
        with open(path) as f:
            src = f.readlines()
        new = src
        for _ in range(4):                                  # (a wrapped line can expose another statement split: repeat until nothing moves)
            nxt = [l + "\n" for l in reformat(new, limit)]
            if nxt == new:
                break
            new = nxt
        if new != src:
            changed += 1
            if check:
                print("would change", path)
            else:
                with open(path, "w") as f:
                    f.writelines(new)
                print("%s: %d -> %d lines, %d still over %d columns" % (path, len(src), len(new), sum(1 for l in new if len(l.rstrip()) > limit), limit))
    return 1 if check and changed else 0


if __name__ == "__main__":
    sys.exit(main())
