"""ds2hip: ctypes binding (lib) and tensor-level wrappers (ops) of libds2hip.so."""
