def CompileAdaptive(fun):   # eager: no graph compiler in the golden run
    return fun


def CompileTF(fun):
    return fun
