"""tf.app.flags stand-in: `--name=value` / `--name value` command-line flags with
the reference's names and defaults (sse_train.py:60-88, sse_index.py:47-49,
sse_demo.py:48-50).  Unknown positional arguments are kept in `.rest`."""


class _Values(object):
    pass


class FlagSet(object):
    def __init__(self, prog, specs):
        self.prog = prog
        self.specs = specs                      # (name, type, default, help)

    def parse(self, argv):
        out = _Values()
        types = {}
        for name, typ, default, _ in self.specs:
            setattr(out, name, default)
            types[name] = typ
        rest, i = [], 0
        while i < len(argv):
            a = argv[i]
            if a.startswith("--"):
                body = a[2:]
                if body in ("help", "h"):
                    print(self.usage())
                    raise SystemExit(0)
                if "=" in body:
                    name, val = body.split("=", 1)
                else:
                    name = body
                    if types.get(name) is bool and (i + 1 >= len(argv) or argv[i + 1].startswith("--")):
                        val = "true"
                    else:
                        i += 1
                        if i >= len(argv):
                            raise SystemExit("%s: flag --%s needs a value" % (self.prog, name))
                        val = argv[i]
                if name not in types:
                    raise SystemExit("%s: unknown flag --%s\n%s" % (self.prog, name, self.usage()))
                typ = types[name]
                setattr(out, name, (val.lower() in ("1", "true", "yes")) if typ is bool else typ(val))
            else:
                rest.append(a)
            i += 1
        out.rest = rest
        return out

    def usage(self):
        lines = ["usage: %s [--flag=value ...]" % self.prog]
        for name, typ, default, hlp in self.specs:
            lines.append("  --%s (%s, default %r): %s" % (name, typ.__name__, default, hlp))
        return "\n".join(lines)
