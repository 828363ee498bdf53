/* A stand-in for the two value classes of the `cv2` wheel (absent from the image) whose COST is like the wheel's:
 * KeyPoint and DMatch are C structs behind a Python object, built eagerly from their constructor arguments, `pt`
 * makes a new tuple on every read, every field is writable, and KeyPoint_convert crosses the list in one C pass.
 * Test and bench infrastructure only (tests/cv2_stub.py installs it as `cv2` in a child interpreter); the product
 * never imports it.  What it mirrors (OpenCV 4.x Python bindings as published):
 *
 *   cv2.KeyPoint()                                  pt (0, 0), size 0, angle -1, response 0, octave 0, class_id -1
 *   cv2.KeyPoint(x, y, size[, angle[, response[, octave[, class_id]]]])
 *   cv2.DMatch()                                    queryIdx -1, trainIdx -1, imgIdx -1, distance FLT_MAX
 *   cv2.DMatch(q, t, distance) / cv2.DMatch(q, t, imgIdx, distance)
 *   cv2.KeyPoint_convert(keypoints) -> points2f     (here: _kp_to_xy(keypoints, out_buffer); the numpy allocation is python's)
 *   cv2.KeyPoint_convert(points2f[, size[, response[, octave[, class_id]]]]) -> tuple of KeyPoint   (response defaults to 1)
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>
#include <float.h>

typedef struct {
    PyObject_HEAD
    float x, y, size, angle, response;
    int octave, class_id;
} KeyPointObject;

typedef struct {
    PyObject_HEAD
    int queryIdx, trainIdx, imgIdx;
    float distance;
} DMatchObject;

static PyTypeObject KeyPointType, DMatchType;

/* ------------------------------------------------------------------ KeyPoint */
static int keypoint_init(KeyPointObject *self, PyObject *args, PyObject *kw)
{
    static char *names[] = {"x", "y", "size", "angle", "response", "octave", "class_id", NULL};
    float x = 0, y = 0, size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    if (PyTuple_GET_SIZE(args) == 0 && (kw == NULL || PyDict_Size(kw) == 0)) {
        self->x = self->y = self->size = 0; self->angle = -1; self->response = 0; self->octave = 0; self->class_id = -1;
        return 0;
    }
    if (!PyArg_ParseTupleAndKeywords(args, kw, "fff|ffii:KeyPoint", names, &x, &y, &size, &angle, &response, &octave, &class_id))
        return -1;
    self->x = x; self->y = y; self->size = size; self->angle = angle; self->response = response;
    self->octave = octave; self->class_id = class_id;
    return 0;
}

static PyObject *keypoint_get_pt(KeyPointObject *self, void *unused)
{
    (void)unused;
    return Py_BuildValue("(dd)", (double)self->x, (double)self->y);
}

static int keypoint_set_pt(KeyPointObject *self, PyObject *v, void *unused)
{
    (void)unused;
    float x, y;
    if (v == NULL) { PyErr_SetString(PyExc_TypeError, "Cannot delete the pt attribute"); return -1; }
    PyObject *t = PySequence_Tuple(v);
    if (t == NULL) return -1;
    int ok = PyArg_ParseTuple(t, "ff", &x, &y);
    Py_DECREF(t);
    if (!ok) return -1;
    self->x = x; self->y = y;
    return 0;
}

static PyGetSetDef keypoint_getset[] = {
    {"pt", (getter)keypoint_get_pt, (setter)keypoint_set_pt, "coordinates", NULL},
    {NULL, NULL, NULL, NULL, NULL}
};

static PyMemberDef keypoint_members[] = {
    {"size", T_FLOAT, offsetof(KeyPointObject, size), 0, NULL},
    {"angle", T_FLOAT, offsetof(KeyPointObject, angle), 0, NULL},
    {"response", T_FLOAT, offsetof(KeyPointObject, response), 0, NULL},
    {"octave", T_INT, offsetof(KeyPointObject, octave), 0, NULL},
    {"class_id", T_INT, offsetof(KeyPointObject, class_id), 0, NULL},
    {NULL, 0, 0, 0, NULL}
};

static PyObject *keypoint_repr(KeyPointObject *self)
{
    char buf[96];
    snprintf(buf, sizeof buf, "< cv2.KeyPoint (stand-in) (%g, %g) >", self->x, self->y);
    return PyUnicode_FromString(buf);
}

/* ------------------------------------------------------------------ DMatch */
static int dmatch_init(DMatchObject *self, PyObject *args, PyObject *kw)
{
    Py_ssize_t n = PyTuple_GET_SIZE(args);
    if (kw != NULL && PyDict_Size(kw) != 0) {
        static char *names[] = {"_queryIdx", "_trainIdx", "_imgIdx", "_distance", NULL};
        int q, t, im; float d;
        if (!PyArg_ParseTupleAndKeywords(args, kw, "iiif:DMatch", names, &q, &t, &im, &d)) return -1;
        self->queryIdx = q; self->trainIdx = t; self->imgIdx = im; self->distance = d;
        return 0;
    }
    if (n == 0) {
        self->queryIdx = self->trainIdx = self->imgIdx = -1; self->distance = FLT_MAX;
        return 0;
    }
    if (n == 3) {
        int q, t; float d;
        if (!PyArg_ParseTuple(args, "iif:DMatch", &q, &t, &d)) return -1;
        self->queryIdx = q; self->trainIdx = t; self->imgIdx = -1; self->distance = d;
        return 0;
    }
    {
        int q, t, im; float d;
        if (!PyArg_ParseTuple(args, "iiif:DMatch", &q, &t, &im, &d)) return -1;
        self->queryIdx = q; self->trainIdx = t; self->imgIdx = im; self->distance = d;
    }
    return 0;
}

static PyMemberDef dmatch_members[] = {
    {"queryIdx", T_INT, offsetof(DMatchObject, queryIdx), 0, NULL},
    {"trainIdx", T_INT, offsetof(DMatchObject, trainIdx), 0, NULL},
    {"imgIdx", T_INT, offsetof(DMatchObject, imgIdx), 0, NULL},
    {"distance", T_FLOAT, offsetof(DMatchObject, distance), 0, NULL},
    {NULL, 0, 0, 0, NULL}
};

static PyObject *dmatch_repr(DMatchObject *self)
{
    char buf[96];
    snprintf(buf, sizeof buf, "< cv2.DMatch (stand-in) %d -> %d >", self->queryIdx, self->trainIdx);
    return PyUnicode_FromString(buf);
}

/* ------------------------------------------------------------------ KeyPoint_convert */
/* _kp_to_xy(keypoints, out): `out` is a writable buffer of 2 * len(keypoints) floats */
static PyObject *kp_to_xy(PyObject *mod, PyObject *args)
{
    (void)mod;
    PyObject *seq, *fast;
    Py_buffer out;
    if (!PyArg_ParseTuple(args, "Ow*:_kp_to_xy", &seq, &out)) return NULL;
    fast = PySequence_Fast(seq, "keypoints: a sequence of cv2.KeyPoint is expected");
    if (fast == NULL) { PyBuffer_Release(&out); return NULL; }
    Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    if (out.len != (Py_ssize_t)(n * 2 * sizeof(float))) {
        PyErr_SetString(PyExc_ValueError, "_kp_to_xy: output buffer size");
        goto fail;
    }
    float *p = (float *)out.buf;
    PyObject **items = PySequence_Fast_ITEMS(fast);
    for (Py_ssize_t i = 0; i < n; i++) {
        if (!PyObject_TypeCheck(items[i], &KeyPointType)) {
            PyErr_SetString(PyExc_TypeError, "Expected cv::KeyPoint for argument 'keypoints'");     /* cv2 raises cv2.error / TypeError */
            goto fail;
        }
        KeyPointObject *k = (KeyPointObject *)items[i];
        p[2 * i] = k->x; p[2 * i + 1] = k->y;
    }
    Py_DECREF(fast); PyBuffer_Release(&out);
    Py_RETURN_NONE;
fail:
    Py_DECREF(fast); PyBuffer_Release(&out);
    return NULL;
}

/* _xy_to_kp(points2f buffer (n x 2 float32, contiguous), size, response, octave, class_id) -> tuple of KeyPoint */
static PyObject *xy_to_kp(PyObject *mod, PyObject *args)
{
    (void)mod;
    Py_buffer in;
    float size = 1, response = 1;
    int octave = 0, class_id = -1;
    if (!PyArg_ParseTuple(args, "y*|ffii:_xy_to_kp", &in, &size, &response, &octave, &class_id)) return NULL;
    if (in.len % (2 * sizeof(float)) != 0) {
        PyBuffer_Release(&in);
        PyErr_SetString(PyExc_ValueError, "_xy_to_kp: buffer is not n x 2 float32");
        return NULL;
    }
    Py_ssize_t n = in.len / (2 * sizeof(float));
    const float *p = (const float *)in.buf;
    PyObject *out = PyTuple_New(n);
    if (out == NULL) { PyBuffer_Release(&in); return NULL; }
    for (Py_ssize_t i = 0; i < n; i++) {
        KeyPointObject *k = PyObject_New(KeyPointObject, &KeyPointType);
        if (k == NULL) { Py_DECREF(out); PyBuffer_Release(&in); return NULL; }
        k->x = p[2 * i]; k->y = p[2 * i + 1]; k->size = size; k->angle = -1; k->response = response;
        k->octave = octave; k->class_id = class_id;
        PyTuple_SET_ITEM(out, i, (PyObject *)k);
    }
    PyBuffer_Release(&in);
    return out;
}

static PyMethodDef module_methods[] = {
    {"_kp_to_xy", kp_to_xy, METH_VARARGS, "keypoints -> xy into a caller's buffer"},
    {"_xy_to_kp", xy_to_kp, METH_VARARGS, "n x 2 float32 buffer -> tuple of KeyPoint"},
    {NULL, NULL, 0, NULL}
};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_cv2like", "cv2-like KeyPoint / DMatch value classes (test stand-in)", -1,
                                       module_methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__cv2like(void)
{
    KeyPointType = (PyTypeObject){PyVarObject_HEAD_INIT(NULL, 0)};
    KeyPointType.tp_name = "cv2.KeyPoint";
    KeyPointType.tp_basicsize = sizeof(KeyPointObject);
    KeyPointType.tp_flags = Py_TPFLAGS_DEFAULT;
    KeyPointType.tp_new = PyType_GenericNew;
    KeyPointType.tp_init = (initproc)keypoint_init;
    KeyPointType.tp_getset = keypoint_getset;
    KeyPointType.tp_members = keypoint_members;
    KeyPointType.tp_repr = (reprfunc)keypoint_repr;
    DMatchType = (PyTypeObject){PyVarObject_HEAD_INIT(NULL, 0)};
    DMatchType.tp_name = "cv2.DMatch";
    DMatchType.tp_basicsize = sizeof(DMatchObject);
    DMatchType.tp_flags = Py_TPFLAGS_DEFAULT;
    DMatchType.tp_new = PyType_GenericNew;
    DMatchType.tp_init = (initproc)dmatch_init;
    DMatchType.tp_members = dmatch_members;
    DMatchType.tp_repr = (reprfunc)dmatch_repr;
    if (PyType_Ready(&KeyPointType) < 0 || PyType_Ready(&DMatchType) < 0) return NULL;
    PyObject *m = PyModule_Create(&moduledef);
    if (m == NULL) return NULL;
    Py_INCREF(&KeyPointType); Py_INCREF(&DMatchType);
    PyModule_AddObject(m, "KeyPoint", (PyObject *)&KeyPointType);
    PyModule_AddObject(m, "DMatch", (PyObject *)&DMatchType);
    return m;
}
