/*
 * xform.h — host-side 4x4 transform construction for scene objects.
 *
 * Role of utilityCore::buildTransformationMatrix (src/utilities.cpp:85-93) and of
 * the glm::inverse / glm::inverseTranspose calls in Scene::loadFromJSON
 * (src/scene.cpp:848-851).  Geometry tests run on these matrices and hit
 * results are compared bit for bit, so the floating-point operation order of
 * GLM 0.9.6.3 is kept: translate/rotate/scale as in gtc/matrix_transform.inl:40-134,
 * the product as in detail/type_mat4x4.inl (column sums left to right), the
 * inverse as in type_mat4x4.inl:37-92 and the inverse transpose as in
 * gtc/matrix_inverse.inl:95-147.
 */
#pragma once
#include <cmath>
#include <cstring>

namespace ptx {

struct Vec4 { float v[4]; };
struct Mat4 { float c[4][4]; }; /* c[column][row] */

inline Vec4 mul(const float* col, float s) { return Vec4{ { col[0] * s, col[1] * s, col[2] * s, col[3] * s } }; }
inline Vec4 add(Vec4 a, Vec4 b) { return Vec4{ { a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3] } }; }
inline Vec4 sub(Vec4 a, Vec4 b) { return Vec4{ { a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3] } }; }
inline Vec4 mulv(Vec4 a, Vec4 b) { return Vec4{ { a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2], a.v[3] * b.v[3] } }; }
inline void setCol(Mat4& m, int j, Vec4 a) { memcpy(m.c[j], a.v, 16); }

inline Mat4 identity()
{
    Mat4 m;
    memset(&m, 0, sizeof(m));
    m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.0f;
    return m;
}

inline Mat4 translate(const Mat4& m, const float v[3])
{
    Mat4 r = m;
    Vec4 t = add(add(add(mul(m.c[0], v[0]), mul(m.c[1], v[1])), mul(m.c[2], v[2])), Vec4{ { m.c[3][0], m.c[3][1], m.c[3][2], m.c[3][3] } });
    setCol(r, 3, t);
    return r;
}

inline Mat4 rotate(const Mat4& m, float angle, const float v[3])
{
    const float a = angle;
    const float c = cosf(a);
    const float s = sinf(a);
    float d = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    float inv = 1.0f / sqrtf(d);
    float axis[3] = { v[0] * inv, v[1] * inv, v[2] * inv };
    float temp[3] = { (1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2] };

    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
    R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
    R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
    R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
    R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];

    Mat4 r;
    for (int j = 0; j < 3; ++j)
        setCol(r, j, add(add(mul(m.c[0], R[j][0]), mul(m.c[1], R[j][1])), mul(m.c[2], R[j][2])));
    memcpy(r.c[3], m.c[3], 16);
    return r;
}

inline Mat4 scale(const Mat4& m, const float v[3])
{
    Mat4 r;
    setCol(r, 0, mul(m.c[0], v[0]));
    setCol(r, 1, mul(m.c[1], v[1]));
    setCol(r, 2, mul(m.c[2], v[2]));
    memcpy(r.c[3], m.c[3], 16);
    return r;
}

inline Mat4 matmul(const Mat4& A, const Mat4& B)
{
    Mat4 r;
    for (int j = 0; j < 4; ++j)
        setCol(r, j, add(add(add(mul(A.c[0], B.c[j][0]), mul(A.c[1], B.c[j][1])), mul(A.c[2], B.c[j][2])), mul(A.c[3], B.c[j][3])));
    return r;
}

inline Mat4 inverse(const Mat4& M)
{
    const float(*m)[4] = M.c;
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    Vec4 Fac0{ { Coef00, Coef00, Coef02, Coef03 } };
    Vec4 Fac1{ { Coef04, Coef04, Coef06, Coef07 } };
    Vec4 Fac2{ { Coef08, Coef08, Coef10, Coef11 } };
    Vec4 Fac3{ { Coef12, Coef12, Coef14, Coef15 } };
    Vec4 Fac4{ { Coef16, Coef16, Coef18, Coef19 } };
    Vec4 Fac5{ { Coef20, Coef20, Coef22, Coef23 } };

    Vec4 Vec0{ { m[1][0], m[0][0], m[0][0], m[0][0] } };
    Vec4 Vec1{ { m[1][1], m[0][1], m[0][1], m[0][1] } };
    Vec4 Vec2{ { m[1][2], m[0][2], m[0][2], m[0][2] } };
    Vec4 Vec3{ { m[1][3], m[0][3], m[0][3], m[0][3] } };

    Vec4 Inv0 = add(sub(mulv(Vec1, Fac0), mulv(Vec2, Fac1)), mulv(Vec3, Fac2));
    Vec4 Inv1 = add(sub(mulv(Vec0, Fac0), mulv(Vec2, Fac3)), mulv(Vec3, Fac4));
    Vec4 Inv2 = add(sub(mulv(Vec0, Fac1), mulv(Vec1, Fac3)), mulv(Vec3, Fac5));
    Vec4 Inv3 = add(sub(mulv(Vec0, Fac2), mulv(Vec1, Fac4)), mulv(Vec2, Fac5));

    Vec4 SignA{ { +1, -1, +1, -1 } };
    Vec4 SignB{ { -1, +1, -1, +1 } };
    Mat4 Inv;
    setCol(Inv, 0, mulv(Inv0, SignA));
    setCol(Inv, 1, mulv(Inv1, SignB));
    setCol(Inv, 2, mulv(Inv2, SignA));
    setCol(Inv, 3, mulv(Inv3, SignB));

    Vec4 Row0{ { Inv.c[0][0], Inv.c[1][0], Inv.c[2][0], Inv.c[3][0] } };
    Vec4 Dot0 = mulv(Vec4{ { m[0][0], m[0][1], m[0][2], m[0][3] } }, Row0);
    float Dot1 = (Dot0.v[0] + Dot0.v[1]) + (Dot0.v[2] + Dot0.v[3]);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) Inv.c[j][i] = Inv.c[j][i] * OneOverDeterminant;
    return Inv;
}

inline Mat4 inverseTranspose(const Mat4& M)
{
    const float(*m)[4] = M.c;
    float SubFactor00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float SubFactor01 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float SubFactor02 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float SubFactor03 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float SubFactor04 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float SubFactor05 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float SubFactor06 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float SubFactor07 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float SubFactor08 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float SubFactor09 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float SubFactor10 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float SubFactor11 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float SubFactor12 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float SubFactor13 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float SubFactor14 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float SubFactor15 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float SubFactor16 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float SubFactor17 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float SubFactor18 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    Mat4 I;
    I.c[0][0] = +(m[1][1] * SubFactor00 - m[1][2] * SubFactor01 + m[1][3] * SubFactor02);
    I.c[0][1] = -(m[1][0] * SubFactor00 - m[1][2] * SubFactor03 + m[1][3] * SubFactor04);
    I.c[0][2] = +(m[1][0] * SubFactor01 - m[1][1] * SubFactor03 + m[1][3] * SubFactor05);
    I.c[0][3] = -(m[1][0] * SubFactor02 - m[1][1] * SubFactor04 + m[1][2] * SubFactor05);

    I.c[1][0] = -(m[0][1] * SubFactor00 - m[0][2] * SubFactor01 + m[0][3] * SubFactor02);
    I.c[1][1] = +(m[0][0] * SubFactor00 - m[0][2] * SubFactor03 + m[0][3] * SubFactor04);
    I.c[1][2] = -(m[0][0] * SubFactor01 - m[0][1] * SubFactor03 + m[0][3] * SubFactor05);
    I.c[1][3] = +(m[0][0] * SubFactor02 - m[0][1] * SubFactor04 + m[0][2] * SubFactor05);

    I.c[2][0] = +(m[0][1] * SubFactor06 - m[0][2] * SubFactor07 + m[0][3] * SubFactor08);
    I.c[2][1] = -(m[0][0] * SubFactor06 - m[0][2] * SubFactor09 + m[0][3] * SubFactor10);
    I.c[2][2] = +(m[0][0] * SubFactor11 - m[0][1] * SubFactor09 + m[0][3] * SubFactor12);
    I.c[2][3] = -(m[0][0] * SubFactor08 - m[0][1] * SubFactor10 + m[0][2] * SubFactor12);

    I.c[3][0] = -(m[0][1] * SubFactor13 - m[0][2] * SubFactor14 + m[0][3] * SubFactor15);
    I.c[3][1] = +(m[0][0] * SubFactor13 - m[0][2] * SubFactor16 + m[0][3] * SubFactor17);
    I.c[3][2] = -(m[0][0] * SubFactor14 - m[0][1] * SubFactor16 + m[0][3] * SubFactor18);
    I.c[3][3] = +(m[0][0] * SubFactor15 - m[0][1] * SubFactor17 + m[0][2] * SubFactor18);

    float Determinant = +m[0][0] * I.c[0][0] + m[0][1] * I.c[0][1] + m[0][2] * I.c[0][2] + m[0][3] * I.c[0][3];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) I.c[j][i] = I.c[j][i] / Determinant;
    return I;
}

/* T * Rx * Ry * Rz * S with angles in degrees */
inline Mat4 buildTransformationMatrix(const float t[3], const float rot[3], const float s[3])
{
    const float PI = 3.1415926535897932384626422832795028841971f;
    const float X[3] = { 1, 0, 0 }, Y[3] = { 0, 1, 0 }, Z[3] = { 0, 0, 1 };
    Mat4 I = identity();
    Mat4 T = translate(I, t);
    Mat4 R = rotate(I, rot[0] * PI / 180, X);
    R = matmul(R, rotate(I, rot[1] * PI / 180, Y));
    R = matmul(R, rotate(I, rot[2] * PI / 180, Z));
    Mat4 S = scale(I, s);
    return matmul(matmul(T, R), S);
}

} // namespace ptx
