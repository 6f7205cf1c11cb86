/*
 * _tvbf_records -- the reference's record form of a top-K result, built in C.
 *
 * scripts/populate_database.py:210-221 of the reference hands the database layer a
 * dict[int, list[dict]]: per show a list of
 *   {"similar_show_id", "similarity_score", "genre_score", "text_score", "metadata_score"}.
 * For 100,000 shows x 20 neighbours that is two million small dicts; building them in a Python loop
 * takes 15 steps of the GPU path.  This module builds the same objects from the result arrays
 * directly (one pass, interned keys, the cyclic collector paused for the duration of the call --
 * it would otherwise walk the young dicts again and again), which takes about a third of the time.
 * Host code only: no GPU, no dependency beyond CPython.
 *
 *   records(ids, idx, score, comp, count, k, start, stop) -> dict
 *     ids    bytes-like, int64[*]      show id of every row of the catalogue
 *     idx    bytes-like, int32[n][k]   neighbour rows, -1 in unused slots
 *     score  bytes-like, float64[n][k]
 *     comp   bytes-like, float64[n][k][3]  genre / text / metadata
 *     count  bytes-like, int32[n]      neighbours kept per row
 *   rows start <= r < stop with count[r] > 0 appear in the result, keyed by ids[r].
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *k_sid, *k_sim, *k_genre, *k_text, *k_meta;

static int set_steal(PyObject* d, PyObject* key, PyObject* value) {  /* takes over `value` */
    int rc;
    if (!value) return -1;
    rc = PyDict_SetItem(d, key, value);
    Py_DECREF(value);
    return rc;
}

static PyObject* records(PyObject* self, PyObject* args) {
    Py_buffer ids, idx, score, comp, count;
    Py_ssize_t k, start, stop, n, n_ids, r;
    PyObject* out = NULL;
    int gc_was_on, failed = 0;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*y*y*y*y*nnn", &ids, &idx, &score, &comp, &count, &k, &start, &stop)) return NULL;
    n = count.len / (Py_ssize_t)sizeof(int32_t);
    n_ids = ids.len / (Py_ssize_t)sizeof(int64_t);
    if (k < 1 || start < 0 || stop > n || idx.len != n * k * (Py_ssize_t)sizeof(int32_t) ||
        score.len != n * k * (Py_ssize_t)sizeof(double) || comp.len != n * k * 3 * (Py_ssize_t)sizeof(double) || n_ids < stop) {
        PyErr_SetString(PyExc_ValueError, "records(): array sizes do not fit n, k, start, stop");
        goto done;
    }
    out = PyDict_New();
    if (!out) goto done;
    gc_was_on = PyGC_Disable();
    {
        const int64_t* id = (const int64_t*)ids.buf;
        const int32_t* ix = (const int32_t*)idx.buf;
        const double* sc = (const double*)score.buf;
        const double* cp = (const double*)comp.buf;
        const int32_t* cn = (const int32_t*)count.buf;
        for (r = start; r < stop && !failed; ++r) {
            Py_ssize_t c = cn[r], s;
            PyObject *list, *key;
            if (c <= 0) continue;
            if (c > k) c = k;
            list = PyList_New(c);
            if (!list) { failed = 1; break; }
            for (s = 0; s < c; ++s) {
                const int32_t j = ix[r * k + s];
                const double* p = cp + (r * k + s) * 3;
                PyObject* d = PyDict_New();
                if (!d) { failed = 1; break; }
                PyList_SET_ITEM(list, s, d);  /* the list owns it from here */
                if (j < 0 || j >= n_ids) {
                    PyErr_SetString(PyExc_ValueError, "records(): neighbour index outside the catalogue");
                    failed = 1;
                    break;
                }
                if (set_steal(d, k_sid, PyLong_FromLongLong((long long)id[j])) ||
                    set_steal(d, k_sim, PyFloat_FromDouble(sc[r * k + s])) ||
                    set_steal(d, k_genre, PyFloat_FromDouble(p[0])) ||
                    set_steal(d, k_text, PyFloat_FromDouble(p[1])) ||
                    set_steal(d, k_meta, PyFloat_FromDouble(p[2]))) {
                    failed = 1;
                    break;
                }
            }
            if (failed) {
                /* unfilled slots of a new list are NULL; fill them so that the list can be released */
                for (; s < c; ++s)
                    if (!PyList_GET_ITEM(list, s)) { Py_INCREF(Py_None); PyList_SET_ITEM(list, s, Py_None); }
                Py_DECREF(list);
                break;
            }
            key = PyLong_FromLongLong((long long)id[r]);
            if (!key || PyDict_SetItem(out, key, list)) failed = 1;
            Py_XDECREF(key);
            Py_DECREF(list);
        }
    }
    if (gc_was_on) PyGC_Enable();
    if (failed) { Py_CLEAR(out); }
done:
    PyBuffer_Release(&ids);
    PyBuffer_Release(&idx);
    PyBuffer_Release(&score);
    PyBuffer_Release(&comp);
    PyBuffer_Release(&count);
    return out;
}

static PyMethodDef methods[] = {
    {"records", records, METH_VARARGS, "records(ids, idx, score, comp, count, k, start, stop) -> dict[int, list[dict]]"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_tvbf_records", "record dicts of a top-K result", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__tvbf_records(void) {
    k_sid = PyUnicode_InternFromString("similar_show_id");
    k_sim = PyUnicode_InternFromString("similarity_score");
    k_genre = PyUnicode_InternFromString("genre_score");
    k_text = PyUnicode_InternFromString("text_score");
    k_meta = PyUnicode_InternFromString("metadata_score");
    if (!k_sid || !k_sim || !k_genre || !k_text || !k_meta) return NULL;
    return PyModule_Create(&module);
}
