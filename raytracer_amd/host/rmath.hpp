// Host-side math used while CONSTRUCTING scenes (never inside the render path,
// which runs on the device).  Mirrors raytracer/src/math/{vector,matrix}.rs so
// that mesh transforms produce the same f64 values the reference would.
#pragma once
#include <cmath>
#include <cstdint>

namespace rth {

struct Vector3 {
    double x = 0., y = 0., z = 0.;
    Vector3() = default;
    Vector3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    double axis(int a) const { return a == 0 ? x : a == 1 ? y : z; }   // vector.rs:102-108
};
inline Vector3 operator+(Vector3 a, Vector3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector3 operator-(Vector3 a, Vector3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vector3 operator*(Vector3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vector3 operator/(Vector3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline bool operator!=(Vector3 a, Vector3 b) { return a.x != b.x || a.y != b.y || a.z != b.z; }
inline Vector3 cross(Vector3 a, Vector3 b) {                            // vector.rs:65-71
    return {a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x};
}
inline double magnitude(Vector3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // vector.rs:74-76
inline Vector3 normalize(Vector3 a) { return a / magnitude(a); }        // vector.rs:79-81

// matrix.rs:9-22: rotation 3x3 + position
struct Matrix {
    double x = 0., y = 0., z = 0.;
    double m00 = 1., m01 = 0., m02 = 0., m10 = 0., m11 = 1., m12 = 0., m20 = 0., m21 = 0., m22 = 1.;
    static Matrix from_pos(Vector3 v) { Matrix m; m.x = v.x; m.y = v.y; m.z = v.z; return m; }   // matrix.rs:142-146
    Vector3 pos() const { return {x, y, z}; }
};
// matrix.rs:148-185: 4x4 product, every element accumulated from 0.0 in k order
inline Matrix operator*(const Matrix& A, const Matrix& B) {
    double a[4][4] = {{A.m00, A.m01, A.m02, A.x}, {A.m10, A.m11, A.m12, A.y}, {A.m20, A.m21, A.m22, A.z}, {0., 0., 0., 1.}};
    double b[4][4] = {{B.m00, B.m01, B.m02, B.x}, {B.m10, B.m11, B.m12, B.y}, {B.m20, B.m21, B.m22, B.z}, {0., 0., 0., 1.}};
    double o[4][4] = {{0., 0., 0., 0.}, {0., 0., 0., 0.}, {0., 0., 0., 0.}, {0., 0., 0., 0.}};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) o[i][j] += a[i][k] * b[k][j];
    Matrix r;
    r.x = o[0][3]; r.y = o[1][3]; r.z = o[2][3];
    r.m00 = o[0][0]; r.m01 = o[0][1]; r.m02 = o[0][2];
    r.m10 = o[1][0]; r.m11 = o[1][1]; r.m12 = o[1][2];
    r.m20 = o[2][0]; r.m21 = o[2][1]; r.m22 = o[2][2];
    return r;
}
// matrix.rs:60-66
inline void euler_matrices(double x, double y, double z, Matrix& a, Matrix& b, Matrix& c) {
    // one sincos() per angle: what an optimising compiler makes of a sin/cos pair anyway, said outright so that the
    // result does not depend on the compiler (glibc's sincos and sin can differ in the last bit)
    double sx, cx, sy, cy, sz, cz;
    ::sincos(x, &sx, &cx); ::sincos(y, &sy, &cy); ::sincos(z, &sz, &cz);
    a = Matrix(); a.m11 = cx; a.m12 = -sx; a.m21 = sx; a.m22 = cx;
    b = Matrix(); b.m00 = cy; b.m02 = sy; b.m20 = -sy; b.m22 = cy;
    c = Matrix(); c.m00 = cz; c.m01 = -sz; c.m10 = sz; c.m11 = cz;
}
inline Matrix from_euler_xyz(double x, double y, double z) { Matrix a, b, c; euler_matrices(x, y, z, a, b, c); return a * b * c; }  // :69-72
inline Matrix from_euler_zyx(double x, double y, double z) { Matrix a, b, c; euler_matrices(x, y, z, a, b, c); return c * b * a; }  // :75-78

}  // namespace rth
