/* thrifty_amd._fastresults: the per-block result objects of `for detected, result in Detector(...)`
 * (reference thrifty/detect.py:60-91, toads_data.py:22-61) built a BATCH at a time from the engine's
 * thr_record array, in C.
 *
 * The reference's loop hands out one `(detected, DetectionResult)` per block, the result holding two
 * namedtuples (CarrierSyncInfo, CorrDetectionInfo) and five scalars -- about a dozen Python objects,
 * 1.6 us of interpreter time per block when built in Python: 0.6 M blocks/s behind an engine that
 * computes 18 M.  Here a result is ONE object that keeps the block's 64-byte record and makes each
 * attribute when it is first read (and keeps it: reading twice gives the same object; assigning
 * replaces it) -- the values and TYPES are the reference's:
 *   timestamp      the object the block source supplied, untouched
 *   block          int
 *   soa            None without a carrier; new_len * block + sample + offset -- a float when the
 *                  correlation peak was detected, an int when not (offset is the int 0 then,
 *                  soa_estimator.py:88)
 *   carrier_info   CarrierSyncInfo(bin int, offset, energy np.float32, noise np.float32); offset is
 *                  offset_type(value) -- float, np.float32 for PreshiftDetector -- or the int 0 (no
 *                  carrier / THR_FLAG_INT_OFFSET)
 *   corr_info      None without a carrier; CorrDetectionInfo(sample int, offset float | int 0, energy
 *                  float, noise float)
 *   rxid, txid     the detector's rxid; the template index for a several-template detector, else None
 * `ResultBase` is the C base of toads_data.DetectionResult (which adds serialize / deserialize): a
 * result built in Python through the reference's constructor stores the seven values as given.
 *
 * Host glue, no device code: compiled with the C compiler (thrifty_amd/build.py), no numpy headers
 * (np.float32 is called, once per attribute read).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

typedef struct {            /* thr_record, include/thrifty_hip.h */
    int64_t block_idx;
    uint32_t flags;
    int32_t template_id;
    int32_t carrier_bin;
    int32_t corr_sample;
    double carrier_offset;
    double corr_offset;
    float carrier_energy, carrier_noise, corr_energy, corr_noise;
    uint64_t reserved;
} rec_t;

#define FLAG_CARRIER 1u
#define FLAG_CORR 2u
#define FLAG_INT_OFFSET 8u

enum { F_TIMESTAMP, F_BLOCK, F_SOA, F_CARRIER, F_CORR, F_RXID, F_TXID, F_COUNT };
static const char *const FIELD_NAMES[F_COUNT] = {"timestamp", "block",   "soa", "carrier_info",
                                                 "corr_info", "rxid",    "txid"};

/* ------------------------------------------------------------------ context */
typedef struct {
    PyObject_HEAD
    long long new_len;
    PyObject *rxid;
    PyObject *offset_type;   /* float, int or np.float32 */
    PyObject *f32;           /* np.float32 */
    PyTypeObject *car_cls, *cor_cls, *res_cls;
    int multi;
    /* thr_format_toad of libthriftyhip.so (its address, handed over by the caller: this module links
     * nothing) and the carrier-offset mode it takes for offset_type: serialize() of an untouched,
     * detected result is then the library's text for its record -- the text thr_run_card writes */
    void *format_fn;
    int offset_mode;
} Ctx;

typedef int (*format_toad_fn)(const void *recs, const double *timestamps, size_t n, int64_t new_len, int with_rxid,
                              int64_t rxid, int with_txid, int carrier_offset_f32, char *out, size_t out_capacity,
                              size_t *out_used);

static void ctx_dealloc(Ctx *c) {
    Py_XDECREF(c->rxid);
    Py_XDECREF(c->offset_type);
    Py_XDECREF(c->f32);
    Py_XDECREF((PyObject *)c->car_cls);
    Py_XDECREF((PyObject *)c->cor_cls);
    Py_XDECREF((PyObject *)c->res_cls);
    Py_TYPE(c)->tp_free((PyObject *)c);
}

static PyTypeObject ResultType;

static int ctx_init(Ctx *c, PyObject *args, PyObject *kw) {
    static char *names[] = {"new_len", "rxid", "offset_type", "float32", "carrier_cls", "corr_cls",
                            "result_cls", "multi", "format_fn", "offset_mode", NULL};
    PyObject *rxid, *ot, *f32, *car, *cor, *res;
    long long new_len;
    int multi = 0, offset_mode = 0;
    unsigned long long format_fn = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "LOOOOOO|pKi", names, &new_len, &rxid, &ot, &f32, &car, &cor,
                                     &res, &multi, &format_fn, &offset_mode))
        return -1;
    if (!PyType_Check(car) || !PyType_IsSubtype((PyTypeObject *)car, &PyTuple_Type) || !PyType_Check(cor) ||
        !PyType_IsSubtype((PyTypeObject *)cor, &PyTuple_Type)) {
        PyErr_SetString(PyExc_TypeError, "carrier_cls / corr_cls must be tuple subclasses (namedtuples)");
        return -1;
    }
    if (!PyType_Check(res) || !PyType_IsSubtype((PyTypeObject *)res, &ResultType) ||
        ((PyTypeObject *)res)->tp_basicsize != ResultType.tp_basicsize) {
        PyErr_SetString(PyExc_TypeError, "result_cls must be a ResultBase subclass without instance storage of its own");
        return -1;
    }
    if (!PyCallable_Check(ot) || !PyCallable_Check(f32)) {
        PyErr_SetString(PyExc_TypeError, "offset_type / float32 must be callable");
        return -1;
    }
    c->new_len = new_len;
    c->multi = multi;
    c->format_fn = (void *)(uintptr_t)format_fn;
    c->offset_mode = offset_mode;
#define TAKE(field, v, T) do { PyObject *old_ = (PyObject *)c->field; Py_INCREF(v); c->field = (T)(v); Py_XDECREF(old_); } while (0)
    TAKE(rxid, rxid, PyObject *);
    TAKE(offset_type, ot, PyObject *);
    TAKE(f32, f32, PyObject *);
    TAKE(car_cls, car, PyTypeObject *);
    TAKE(cor_cls, cor, PyTypeObject *);
    TAKE(res_cls, res, PyTypeObject *);
#undef TAKE
    return 0;
}

static PyTypeObject CtxType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "thrifty_amd._fastresults.Context",
    .tp_basicsize = sizeof(Ctx),
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_doc = "Context(new_len, rxid, offset_type, float32, carrier_cls, corr_cls, result_cls, multi=False): what a "
              "batch of results shares",
    .tp_new = PyType_GenericNew,
    .tp_init = (initproc)ctx_init,
    .tp_dealloc = (destructor)ctx_dealloc,
};

/* ------------------------------------------------------------------ result */
typedef struct {
    PyObject_HEAD
    Ctx *ctx;               /* NULL: built through the Python constructor, every field given */
    rec_t rec;
    PyObject *f[F_COUNT];   /* given / assigned / already made values */
    int touched;            /* an attribute has been assigned: the record no longer says everything */
} Result;

/* A batch's results die together -- the loop drops each as it goes -- and with them every pymalloc
 * arena they filled: the next batch would map fresh pages again, and on a process whose input window
 * is page-locking a file (mlock / munlock hold the address-space lock for milliseconds) those page
 * faults made build() 0.8 us per block instead of 0.1.  So the storage of dead results is kept
 * (at most FREE_CAP objects of exactly this size) and handed to the next batch. */
#define FREE_CAP 8192
static void *free_results[FREE_CAP];
static int n_free_results = 0;

static void result_free(void *p) {
    PyObject *o = (PyObject *)p;
    if (Py_TYPE(o)->tp_basicsize == (Py_ssize_t)sizeof(Result) && Py_TYPE(o)->tp_itemsize == 0 &&
        !PyType_IS_GC(Py_TYPE(o)) && n_free_results < FREE_CAP)
        free_results[n_free_results++] = p;
    else
        PyObject_Free(p);
}

static Result *result_alloc(PyTypeObject *cls) {
    if (n_free_results > 0 && cls->tp_basicsize == (Py_ssize_t)sizeof(Result) && !PyType_IS_GC(cls)) {
        Result *r = (Result *)free_results[--n_free_results];
        memset(r, 0, sizeof(Result));
        PyObject_Init((PyObject *)r, cls);      /* reference count 1, a reference on a heap type */
        return r;
    }
    return (Result *)cls->tp_alloc(cls, 0);     /* zero-filled */
}

static void result_dealloc(Result *r) {
    /* (a Python subclass's instances come here through subtype_dealloc, which drops the reference an
     * instance holds on its heap type itself) */
    for (int i = 0; i < F_COUNT; ++i) Py_XDECREF(r->f[i]);
    Py_XDECREF((PyObject *)r->ctx);
    Py_TYPE(r)->tp_free((PyObject *)r);
}

static int result_init(Result *r, PyObject *args, PyObject *kw) {
    static char *names[] = {"timestamp", "block", "soa", "carrier_info", "corr_info", "rxid", "txid", NULL};
    PyObject *v[F_COUNT] = {NULL, NULL, NULL, NULL, NULL, Py_None, Py_None};
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OOOOO|OO", names, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]))
        return -1;
    for (int i = 0; i < F_COUNT; ++i) {
        PyObject *old = r->f[i];
        Py_INCREF(v[i]);
        r->f[i] = v[i];
        Py_XDECREF(old);
    }
    Py_CLEAR(r->ctx);
    return 0;
}

static PyObject *call1(PyObject *fn, PyObject *arg) {   /* fn(arg), steals arg */
    if (arg == NULL) return NULL;
    PyObject *out = PyObject_CallOneArg(fn, arg);
    Py_DECREF(arg);
    return out;
}

static PyObject *make_tuple4(PyTypeObject *cls, PyObject *a, PyObject *b, PyObject *c, PyObject *d) {
    PyObject *t = NULL;
    if (a && b && c && d) t = cls->tp_alloc(cls, 4);   /* what tuple.__new__(cls, ...) does for a subclass */
    if (t == NULL) {
        Py_XDECREF(a);
        Py_XDECREF(b);
        Py_XDECREF(c);
        Py_XDECREF(d);
        return NULL;
    }
    PyTuple_SET_ITEM(t, 0, a);
    PyTuple_SET_ITEM(t, 1, b);
    PyTuple_SET_ITEM(t, 2, c);
    PyTuple_SET_ITEM(t, 3, d);
    return t;
}

static PyObject *result_make(Result *r, int which) {
    const Ctx *c = r->ctx;
    const rec_t *q = &r->rec;
    if (c == NULL) {
        PyErr_Format(PyExc_AttributeError, "%s", FIELD_NAMES[which]);
        return NULL;
    }
    const int carrier = (q->flags & FLAG_CARRIER) != 0, hit = (q->flags & FLAG_CORR) != 0;
    switch (which) {
        case F_BLOCK:
            return PyLong_FromLongLong(q->block_idx);
        case F_SOA: {
            if (!carrier) Py_RETURN_NONE;
            const long long base = c->new_len * q->block_idx + q->corr_sample;
            return hit ? PyFloat_FromDouble((double)base + q->corr_offset) : PyLong_FromLongLong(base);
        }
        case F_CARRIER: {
            PyObject *off;
            if (!carrier || (q->flags & FLAG_INT_OFFSET))
                off = PyLong_FromLong(0);
            else if (c->offset_type == (PyObject *)&PyFloat_Type)
                off = PyFloat_FromDouble(q->carrier_offset);
            else
                off = call1(c->offset_type, PyFloat_FromDouble(q->carrier_offset));
            return make_tuple4(c->car_cls, PyLong_FromLong(q->carrier_bin), off,
                               call1(c->f32, PyFloat_FromDouble((double)q->carrier_energy)),
                               call1(c->f32, PyFloat_FromDouble((double)q->carrier_noise)));
        }
        case F_CORR:
            if (!carrier) Py_RETURN_NONE;
            return make_tuple4(c->cor_cls, PyLong_FromLong(q->corr_sample),
                               hit ? PyFloat_FromDouble(q->corr_offset) : PyLong_FromLong(0),
                               PyFloat_FromDouble((double)q->corr_energy), PyFloat_FromDouble((double)q->corr_noise));
        case F_RXID:
            Py_INCREF(c->rxid);
            return c->rxid;
        case F_TXID:
            if (c->multi) return PyLong_FromLong(q->template_id);
            Py_RETURN_NONE;
        default:   /* F_TIMESTAMP is always given */
            PyErr_Format(PyExc_AttributeError, "%s", FIELD_NAMES[which]);
            return NULL;
    }
}

static PyObject *result_get(Result *r, void *closure) {
    const int which = (int)(intptr_t)closure;
    if (r->f[which] == NULL) {
        r->f[which] = result_make(r, which);
        if (r->f[which] == NULL) return NULL;
    }
    Py_INCREF(r->f[which]);
    return r->f[which];
}

static int result_set(Result *r, PyObject *value, void *closure) {
    const int which = (int)(intptr_t)closure;
    if (value == NULL) {
        PyErr_Format(PyExc_AttributeError, "cannot delete %s", FIELD_NAMES[which]);
        return -1;
    }
    PyObject *old = r->f[which];
    Py_INCREF(value);
    r->f[which] = value;
    r->touched = 1;
    Py_XDECREF(old);
    return 0;
}

/* _serialize_fast() -> the .toad line of this result as the engine library formats it
 * (thr_format_toad: the text of serialize(), byte for byte -- tests/test_host_logic.py,
 * tests/test_fastresults.py), or None when the record does not say everything: a result built or
 * changed in Python, an undetected block, a timestamp that is not a float / int, an rxid that is
 * not an int / None.  toads_data.DetectionResult.serialize() formats in Python then. */
static PyObject *result_serialize_fast(Result *r, PyObject *noargs) {
    const Ctx *c = r->ctx;
    if (c == NULL || c->format_fn == NULL || r->touched || !(r->rec.flags & FLAG_CORR)) Py_RETURN_NONE;
    PyObject *ts = r->f[F_TIMESTAMP];
    double t;
    if (ts != NULL && PyFloat_CheckExact(ts))
        t = PyFloat_AS_DOUBLE(ts);
    else if (ts != NULL && PyLong_CheckExact(ts)) {
        t = PyLong_AsDouble(ts);
        if (t == -1.0 && PyErr_Occurred()) {
            PyErr_Clear();
            Py_RETURN_NONE;
        }
    } else
        Py_RETURN_NONE;
    int with_rxid = 0;
    long long rxid = 0;
    if (c->rxid != Py_None) {
        if (!PyLong_CheckExact(c->rxid)) Py_RETURN_NONE;
        int overflow = 0;
        rxid = PyLong_AsLongLongAndOverflow(c->rxid, &overflow);
        if (overflow || (rxid == -1 && PyErr_Occurred())) {
            PyErr_Clear();
            Py_RETURN_NONE;
        }
        with_rxid = 1;
    }
    char text[512];
    size_t used = 0;
    const int rc = ((format_toad_fn)c->format_fn)(&r->rec, &t, 1, (int64_t)c->new_len, with_rxid, (int64_t)rxid,
                                                  c->multi, c->offset_mode, text, sizeof(text), &used);
    if (rc != 0 || used == 0 || used > sizeof(text)) Py_RETURN_NONE;
    return PyUnicode_FromStringAndSize(text, (Py_ssize_t)used - 1);   /* without the line end */
}

static PyMethodDef result_methods[] = {
    {"_serialize_fast", (PyCFunction)result_serialize_fast, METH_NOARGS,
     "the engine library's .toad line for an untouched detected result, else None"},
    {NULL, NULL, 0, NULL}};

#define FIELD(i, doc) {(char *)0, (getter)result_get, (setter)result_set, doc, (void *)(intptr_t)(i)}
static PyGetSetDef result_getset[] = {
    FIELD(F_TIMESTAMP, "timestamp of the block (as the block source supplied it)"),
    FIELD(F_BLOCK, "block index"),
    FIELD(F_SOA, "sample of arrival: new_len * block + corr_info.sample + corr_info.offset (None without a carrier)"),
    FIELD(F_CARRIER, "CarrierSyncInfo(bin, offset, energy, noise)"),
    FIELD(F_CORR, "CorrDetectionInfo(sample, offset, energy, noise), None without a carrier"),
    FIELD(F_RXID, "receiver id"),
    FIELD(F_TXID, "transmitter id (the template index of a several-template detector)"),
    {NULL, NULL, NULL, NULL, NULL}};

static PyTypeObject ResultType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "thrifty_amd._fastresults.ResultBase",
    .tp_basicsize = sizeof(Result),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE,
    .tp_doc = "ResultBase(timestamp, block, soa, carrier_info, corr_info, rxid=None, txid=None)",
    .tp_new = PyType_GenericNew,
    .tp_init = (initproc)result_init,
    .tp_dealloc = (destructor)result_dealloc,
    .tp_free = result_free,
    .tp_getset = result_getset,
    .tp_methods = result_methods,
};

/* ------------------------------------------------------------------ build */
/* build(context, records, timestamps) -> [(detected, result), ...]
 * records: C-contiguous buffer of n thr_record (64 bytes each); timestamps: a sequence of n objects. */
static PyObject *fr_build(PyObject *self, PyObject *args) {
    Ctx *ctx;
    Py_buffer buf;
    PyObject *stamps;
    if (!PyArg_ParseTuple(args, "O!y*O", &CtxType, &ctx, &buf, &stamps)) return NULL;
    PyObject *out = NULL, *seq = NULL;
    if (buf.len % (Py_ssize_t)sizeof(rec_t) != 0) {
        PyErr_SetString(PyExc_ValueError, "records: not a whole number of 64-byte records");
        goto done;
    }
    const Py_ssize_t n = buf.len / (Py_ssize_t)sizeof(rec_t);
    seq = PySequence_Fast(stamps, "timestamps must be a sequence");
    if (seq == NULL) goto done;
    if (PySequence_Fast_GET_SIZE(seq) != n) {
        PyErr_Format(PyExc_ValueError, "%zd records but %zd timestamps", n, PySequence_Fast_GET_SIZE(seq));
        goto done;
    }
    out = PyList_New(n);
    if (out == NULL) goto done;
    const rec_t *recs = (const rec_t *)buf.buf;
    PyObject **ts = PySequence_Fast_ITEMS(seq);
    PyTypeObject *cls = ctx->res_cls;
    for (Py_ssize_t i = 0; i < n; ++i) {
        Result *r = result_alloc(cls);   /* zero-filled; no __init__ */
        PyObject *pair = r ? PyTuple_New(2) : NULL;
        if (pair == NULL) {
            Py_XDECREF((PyObject *)r);
            Py_CLEAR(out);
            goto done;
        }
        Py_INCREF((PyObject *)ctx);
        r->ctx = ctx;
        memcpy(&r->rec, &recs[i], sizeof(rec_t));
        Py_INCREF(ts[i]);
        r->f[F_TIMESTAMP] = ts[i];
        PyObject *det = (recs[i].flags & FLAG_CORR) ? Py_True : Py_False;
        Py_INCREF(det);
        PyTuple_SET_ITEM(pair, 0, det);
        PyTuple_SET_ITEM(pair, 1, (PyObject *)r);
        PyList_SET_ITEM(out, i, pair);
    }
done:
    Py_XDECREF(seq);
    PyBuffer_Release(&buf);
    return out;
}

static PyMethodDef fr_methods[] = {
    {"build", fr_build, METH_VARARGS,
     "build(context, records, timestamps) -> [(detected, result), ...] for a C-contiguous buffer of thr_record"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef fr_module = {PyModuleDef_HEAD_INIT, "_fastresults",
                                       "batch construction of thrifty detect's per-block result objects", -1,
                                       fr_methods};

PyMODINIT_FUNC PyInit__fastresults(void) {
    for (int i = 0; i < F_COUNT; ++i) result_getset[i].name = FIELD_NAMES[i];
    if (PyType_Ready(&CtxType) < 0 || PyType_Ready(&ResultType) < 0) return NULL;
    PyObject *m = PyModule_Create(&fr_module);
    if (m == NULL) return NULL;
    Py_INCREF(&CtxType);
    Py_INCREF(&ResultType);
    if (PyModule_AddObject(m, "Context", (PyObject *)&CtxType) < 0 ||
        PyModule_AddObject(m, "ResultBase", (PyObject *)&ResultType) < 0 ||
        PyModule_AddIntConstant(m, "RECORD_BYTES", (long)sizeof(rec_t)) < 0) {
        Py_DECREF(m);
        return NULL;
    }
    return m;
}
