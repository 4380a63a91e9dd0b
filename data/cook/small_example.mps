* This is a comment
NAME    smallExample
OBJSENSE
  MAX
ROWS
  N  obj 
  L  r1
  G  r2
COLUMNS
  x    obj    1   r1  1
  x    r2  2
  y    obj -2.3   r1 -1
  z    obj  0.5
  z    r2    -1
  s    r2    -1
  s    r1     1
RHS
  RIGHT    r1 10.75
  RIGHT    r2  -100
ENDATA
