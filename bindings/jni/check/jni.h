/* Declaration-only SUBSET of the Java Native Interface, for `gcc -fsyntax-only` of ../gpcore_jni.c on hosts without a JDK
 * (tests/test_jni_glue_cpu.py).  NOT a JDK header and never used to build a loadable library: the real build takes
 * $JAVA_HOME/include/jni.h.  Types follow the JNI specification (jint = 32-bit, jlong = 64-bit, jsize = jint, arrays as opaque
 * object pointers); only the function-table entries the glue calls are declared, with the specification's signatures. */
#ifndef GPCORE_CHECK_JNI_H
#define GPCORE_CHECK_JNI_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jobjectArray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jdoubleArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jboolean (*ExceptionCheck)(JNIEnv *env);
    void (*ExceptionClear)(JNIEnv *env);
    void (*DeleteLocalRef)(JNIEnv *env, jobject obj);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    jobject (*GetObjectArrayElement)(JNIEnv *env, jobjectArray array, jsize index);
    jbyteArray (*NewByteArray)(JNIEnv *env, jsize len);
    jintArray (*NewIntArray)(JNIEnv *env, jsize len);
    void (*GetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, jbyte *buf);
    void (*GetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, jint *buf);
    void (*GetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, jdouble *buf);
    void (*SetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, const jbyte *buf);
    void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
    void (*SetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, const jdouble *buf);
};
#endif
