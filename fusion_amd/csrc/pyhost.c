/* pyhost.c -- the two inner loops of the reference-typed boundary (hybrid.py:66-75: list of {'corpus_id', 'score'} dicts), in C.
 *
 * Aggregator.fuse takes and returns the reference's RankedLists.  Walking 4 x 27,942 dicts per query from Python costs ~80 ns per
 * field (operator.itemgetter + np.fromiter) and ~250 ns per rebuilt dict: 25 ms per query, five times the device work of a whole
 * 1024-query batch.  These two functions do the same walks against the CPython C API (libfusion_pyhost.so, loaded with
 * ctypes.PyDLL: the caller holds the GIL).  Host plumbing only -- no arithmetic. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* ids[i] = lst[i][kid], sc[i] = float(lst[i][ksc]) for i < n.  0: done; 1: an id that is not a plain int (or does not fit int64):
 * the caller takes the generic route; 2: not a list of dicts with those keys / a score float() rejects: the caller lets the Python
 * route raise what the reference would. */
int fzh_extract(PyObject* lst, PyObject* kid, PyObject* ksc, int64_t* ids, double* sc, Py_ssize_t n) {
    if (!PyList_CheckExact(lst) || PyList_GET_SIZE(lst) != n) return 2;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* item = PyList_GET_ITEM(lst, i);
        if (!PyDict_CheckExact(item)) return 2;
        PyObject* v = PyDict_GetItemWithError(item, kid);
        PyObject* s = v ? PyDict_GetItemWithError(item, ksc) : NULL;
        if (!v || !s) { PyErr_Clear(); return 2; }
        if (!PyLong_CheckExact(v)) return 1;
        int ovf = 0;
        const long long x = PyLong_AsLongLongAndOverflow(v, &ovf);
        if (ovf) return 1;
        double d;
        if (PyFloat_CheckExact(s)) d = PyFloat_AS_DOUBLE(s);
        else {
            d = PyFloat_AsDouble(s);
            if (d == -1.0 && PyErr_Occurred()) { PyErr_Clear(); return 2; }
        }
        ids[i] = (int64_t)x;
        sc[i] = d;
    }
    return 0;
}

/* [{kid: ids[i], ksc: scores[i]} for i] from two equally long lists (new reference; NULL with a Python error set on failure) */
PyObject* fzh_build(PyObject* kid, PyObject* ksc, PyObject* ids, PyObject* scores) {
    if (!PyList_CheckExact(ids) || !PyList_CheckExact(scores) || PyList_GET_SIZE(ids) != PyList_GET_SIZE(scores)) {
        PyErr_SetString(PyExc_TypeError, "fzh_build: two lists of equal length expected");
        return NULL;
    }
    const Py_ssize_t n = PyList_GET_SIZE(ids);
    PyObject* out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* d = PyDict_New();
        if (!d || PyDict_SetItem(d, kid, PyList_GET_ITEM(ids, i)) < 0 || PyDict_SetItem(d, ksc, PyList_GET_ITEM(scores, i)) < 0) {
            Py_XDECREF(d);
            Py_DECREF(out);
            return NULL;
        }
        PyList_SET_ITEM(out, i, d);
    }
    return out;
}
